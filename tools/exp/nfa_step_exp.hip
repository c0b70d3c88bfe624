/*
 * tools/exp/nfa_step_exp.hip — EXPERIMENT (not part of the product): what does one NFA
 * byte step cost on gfx950 in the segment-per-lane layout of sre_hip_nfa.hip, for
 * different forms of the step?  Same staging (sre_hip_tile.h), synthetic tables.
 *
 *   hipcc -O3 --offload-arch=gfx950 -I sregex_amd/csrc -I include -o gpurun_out/nfa_step_exp tools/exp/nfa_step_exp.hip
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "sre_hip_tile.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Params {
    const uint8_t *data;
    uint64_t       n;
    uint32_t       seg_bytes;
    uint64_t       nsegs;
    const uint64_t *acc;     /* [256] */
    const uint64_t *lut;     /* [8][256] */
    uint64_t       init, self, shiftsrc, match;
    uint64_t       grp[4], grpx[4];
    uint64_t      *out;
};

template <int K>
__device__ inline uint32_t
byte_shl(uint32_t v, uint32_t sh)
{
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(v));
    return r;
}

/*
 * VAR:  0 generic slices (NLUT follow slices, no shift)
 *       1 shift-and: (t << 1) | (t & self) | init | NLUT lookups | NGRP group ops
 * W64:  64-bit masks
 * CLEAN: sample t == 0 every 16 bytes (Pike mode bookkeeping)
 */
template <int VAR, bool W64, int NLUT, int NGRP, bool CLEAN, int MINB>
__global__ __launch_bounds__(256, MINB) void
exp_k(Params P)
{
    typedef typename std::conditional<W64, uint64_t, uint32_t>::type M;
    typedef const __attribute__((address_space(3))) M *lds_m_t;
    constexpr int      TILE = 64, WARM = 128;
    constexpr uint32_t ROWB = 2 * TILE + 16;
    __shared__ __attribute__((aligned(16))) M acc_w[256];
    __shared__ __attribute__((aligned(16))) M lut_w[(NLUT ? NLUT : 1) * 256];
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    RowDesc *rows = reinterpret_cast<RowDesc *>(tile + 256 * ROWB);
    const uint32_t tid = threadIdx.x;
    acc_w[tid] = (M) P.acc[tid];
#pragma unroll
    for (int k = 0; k < NLUT; k++) lut_w[k * 256 + tid] = (M) P.lut[k * 256 + tid];
    const uint32_t acc_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) acc_w;
    const uint32_t lut_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) lut_w;
    const uint32_t sh = W64 ? 3u : 2u;
    const M init = (M) P.init, self = (M) P.self, src = (M) P.shiftsrc, match = (M) P.match;
    M grp[4], grpx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { grp[i] = (M) P.grp[i]; grpx[i] = (M) P.grpx[i]; }

    const uint64_t g = (uint64_t) blockIdx.x * 256 + tid;
    const bool     active = g < P.nsegs;
    const int64_t  seg_a = (int64_t) g * P.seg_bytes;
    RowDesc mine;
    mine.addr = 0; mine.lo = 0; mine.hi16 = -1;
    if (active) {
        mine.addr = (uint64_t) (uintptr_t) P.data + (uint64_t) (seg_a - WARM);
        mine.lo = seg_a >= WARM ? 0 : WARM;
        mine.hi16 = (int32_t) (WARM + P.seg_bytes) - 16;
    }
    rows[tid] = mine;
    M        S = init;
    int64_t  last_clean = -1;
    const uint32_t nrounds = WARM / TILE + P.seg_bytes / TILE;
    const uint32_t lag = (tid >> 5) & 1u;
    uint4    regs[4];
    __syncthreads();
    tile_fetch(regs, rows, tid, 0);
    for (uint32_t s = 0; s <= nrounds; s++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tile_store<8, false>(regs, tile, nullptr, tid, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (s < nrounds) tile_fetch(regs, rows, tid, s + 1);
        if (s < lag || s - lag >= nrounds) continue;
        const uint32_t r = s - lag;
        if (!active) continue;
        const int64_t base = seg_a - WARM + (int64_t) r * TILE;
        if (base < 0) continue;
        uint32_t roww[TILE / 4];
        {
            const uint8_t *srcp = tile + tid * ROWB + (r & 1u) * TILE;
#pragma unroll
            for (int x = 0; x < TILE / 16; x++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(srcp + 16 * x);
                roww[4 * x] = v.x; roww[4 * x + 1] = v.y; roww[4 * x + 2] = v.z; roww[4 * x + 3] = v.w;
            }
        }
        int32_t clean_at = -1;
#pragma unroll
        for (int j = 0; j < TILE; j++) {
            const uint32_t word = roww[j >> 2];
            const uint32_t a = (j & 3) == 0 ? byte_shl<0>(word, sh) : (j & 3) == 1 ? byte_shl<1>(word, sh)
                             : (j & 3) == 2 ? byte_shl<2>(word, sh) : byte_shl<3>(word, sh);
            const M av = *(lds_m_t) (uintptr_t) (acc_base + a);
            const M t = S & av;
            M       nx;
            if (VAR == 0) {
                nx = 0;
#pragma unroll
                for (int k = 0; k < NLUT; k++) {
                    const uint32_t w = k < 4 ? (uint32_t) t : (uint32_t) ((uint64_t) t >> 32);
                    const uint32_t off = (k & 3) == 0 ? byte_shl<0>(w, sh) : (k & 3) == 1 ? byte_shl<1>(w, sh)
                                       : (k & 3) == 2 ? byte_shl<2>(w, sh) : byte_shl<3>(w, sh);
                    nx |= *(lds_m_t) (uintptr_t) (lut_base + k * 256 * (uint32_t) sizeof(M) + off);
                }
            } else {
                nx = (M) ((t & src) << 1) | (t & self) | init;
#pragma unroll
                for (int k = 0; k < NLUT; k++) {
                    /* exception bytes: the LOW bytes of the mask */
                    const uint32_t w = (uint32_t) t;
                    const uint32_t off = (k & 3) == 0 ? byte_shl<0>(w, sh) : (k & 3) == 1 ? byte_shl<1>(w, sh)
                                       : (k & 3) == 2 ? byte_shl<2>(w, sh) : byte_shl<3>(w, sh);
                    nx |= *(lds_m_t) (uintptr_t) (lut_base + k * 256 * (uint32_t) sizeof(M) + off);
                }
#pragma unroll
                for (int k = 0; k < NGRP; k++) nx |= (t & grp[k]) ? grpx[k] : (M) 0;
            }
            S = nx;
            if (CLEAN && (j & 15) == 15 && t == 0) clean_at = j + 1;
        }
        if (S & match) S &= ~match;
        if (CLEAN && clean_at >= 0) last_clean = base + clean_at;
    }
    if (active) {
        P.out[2 * g] = (uint64_t) S;
        P.out[2 * g + 1] = (uint64_t) last_clean;
    }
}

__global__ void
gen_k(uint8_t *d, uint64_t n)
{
    const uint64_t i0 = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (uint64_t i = i0; i < i0 + 16 && i < n; i++) {
        const uint32_t m = (uint32_t) (i % 5);
        d[i] = m == 0 ? 'a' : m == 1 ? 'b' : 'c';
    }
}

typedef void (*kern_t)(Params);
struct Variant { const char *name; kern_t k; int static_lds; };

int
main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (4ull << 30);
    uint8_t *d_data;
    CK(hipMalloc(&d_data, n + 4096));
    hipLaunchKernelGGL(gen_k, dim3((uint32_t) ((n / 16 + 255) / 256)), dim3(256), 0, 0, d_data, n);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> acc(256), lut(8 * 256);
    srand(7);
    for (int i = 0; i < 256; i++) acc[i] = ((uint64_t) rand() << 33) ^ ((uint64_t) rand() << 11) ^ rand();
    for (int i = 0; i < 8 * 256; i++) lut[i] = (((uint64_t) rand() << 33) ^ ((uint64_t) rand() << 11) ^ rand()) & 0x0f0f0f0f0f0f0f0full;
    uint64_t *d_acc, *d_lut, *d_out;
    CK(hipMalloc(&d_acc, 256 * 8));
    CK(hipMalloc(&d_lut, 8 * 256 * 8));
    CK(hipMemcpy(d_acc, acc.data(), 256 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lut, lut.data(), 8 * 256 * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 16 * (1u << 20)));

#define V(var, w64, nlut, ngrp, clean, minb) { #var "," #w64 "," #nlut "," #ngrp "," #clean "," #minb, exp_k<var, w64, nlut, ngrp, clean, minb>, 0 }
    const Variant vars[] = {
        V(0, false, 3, 0, true, 3),      /* the round-2 kernel's shape: 19 threads */
        V(0, true, 8, 0, true, 2),       /* 64 threads, generic */
        V(0, true, 6, 0, true, 2),
        V(1, false, 0, 0, true, 4),      /* pure shift-and, 32 bit */
        V(1, false, 0, 2, true, 4),
        V(1, false, 1, 0, true, 4),
        V(1, false, 1, 2, true, 4),
        V(1, false, 2, 0, true, 4),
        V(1, true, 0, 0, true, 4),       /* 64 bit */
        V(1, true, 0, 2, true, 4),
        V(1, true, 1, 0, true, 4),
        V(1, true, 1, 2, true, 4),
        V(1, true, 2, 0, true, 4),
        V(1, true, 2, 2, true, 4),
        V(1, true, 3, 0, true, 4),
        V(1, false, 1, 0, false, 4),
        V(1, true, 1, 0, false, 4),
        V(1, true, 1, 0, true, 3),
        V(1, true, 1, 0, true, 5),
        V(1, false, 1, 0, true, 5),
        V(1, false, 1, 0, true, 6),
    };
    const size_t dyn = 256 * (2 * 64 + 16) + 256 * 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Variant &v : vars) {
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, v.k, 256, dyn));
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(v.k)));
        /* one round of resident workgroups, 1/64 of the slots spare */
        const uint64_t lanes = (uint64_t) ncu * nb * 256 * 63 / 64;
        uint32_t       seg = (uint32_t) ((n + lanes - 1) / lanes);
        seg = (seg + 255) & ~255u;
        Params P;
        memset(&P, 0, sizeof(P));
        P.data = d_data; P.n = n; P.seg_bytes = seg; P.nsegs = n / seg;
        P.acc = d_acc; P.lut = d_lut; P.out = d_out;
        P.init = 0x0000000100000003ull; P.self = 0x1000100010001000ull; P.shiftsrc = 0x7ffffffe7ffffffeull;
        P.match = 0x8000000080000000ull;
        for (int i = 0; i < 4; i++) { P.grp[i] = 0x00f000f0ull << (4 * i); P.grpx[i] = 0x0100000001000000ull << i; }
        const uint32_t grid = (uint32_t) ((P.nsegs + 255) / 256);
        float best = 1e9f, sum = 0;
        const int reps = 4;
        for (int it = 0; it < reps + 1; it++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), dyn, 0, P);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) { sum += ms; if (ms < best) best = ms; }
        }
        const double bytes = (double) P.nsegs * seg;
        printf("%-26s vgpr %3d lds %6zu wg/cu %d seg %6u grid %6u  avg %.3f ms  best %.3f ms  %.2f TB/s  frac %.3f\n", v.name,
               fa.numRegs, (size_t) fa.sharedSizeBytes + dyn, nb, seg, grid, sum / reps, best, bytes / (sum / reps) * 1e-9,
               bytes / (sum / reps) * 1e-9 / 8.0);
        fflush(stdout);
    }
    return 0;
}
