/*
 * tools/exp/nfa_step_exp2.hip — EXPERIMENT 2 (not part of the product): the shift-and byte
 * step with (a) a staging tile of HALF a line per row (the second half of a fetched line
 * waits in registers), so twice as many workgroups share a CU, and (b) the accept-table
 * reads issued a group of bytes ahead of the dependent chain.
 *
 *   hipcc -O3 --offload-arch=gfx950 -I sregex_amd/csrc -I include -o tools/exp/bin/nfa_step_exp2 tools/exp/nfa_step_exp2.hip
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((aligned(16))) RowDesc {
    uint64_t addr;
    int32_t  lo;
    int32_t  hi16;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;

constexpr uint32_t ROWB = 80;       /* 64 bytes of a row + pad */

/* stage s: the line (s >> 1) of the rows of half (s & 1) of the wave; lanes 4q .. 4q + 3 read 64
 * contiguous bytes: pieces 0 / 2 the FIRST halves of the lines of rows q and q + 16, pieces 1 / 3
 * their SECOND halves */
__device__ inline void
tile2_fetch(uint4 (&regs)[4], const RowDesc *rows, uint32_t tid, uint32_t stage)
{
    const uint32_t wbase = (tid & ~63u) + (stage & 1u) * 32u, lane = tid & 63u;
    const int32_t  line_off = (int32_t) ((stage >> 1) * 128u);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t row = wbase + (lane >> 2) + 16u * (i >> 1), col = (lane & 3u) + 4u * (i & 1u);
        const int32_t  off = line_off + (int32_t) (col * 16);
        const uint4    d = *reinterpret_cast<const uint4 *>(&rows[row]);
        u32x4          v = {0, 0, 0, 0};
        if ((off >= (int32_t) d.z) & (off <= (int32_t) d.w)) {
            const uint64_t a = (((uint64_t) d.y << 32) | d.x) + (uint64_t) (int64_t) off;
            v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_unaligned *>(a);
        }
        regs[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

/* every row of the wave gets 64 fresh bytes: the first half of the line just fetched (rows of
 * half s & 1), the second half of the line fetched one stage earlier (the other rows) */
__device__ inline void
tile2_store(const uint4 (&regs)[4], uint4 (&hold)[2], uint8_t *tile, uint32_t tid, uint32_t stage)
{
    const uint32_t wave = tid & ~63u, lane = tid & 63u, h = stage & 1u;
    const uint32_t rnew = wave + h * 32u + (lane >> 2), rold = wave + (1u - h) * 32u + (lane >> 2);
    const uint32_t col = (lane & 3u) * 16u;
    *reinterpret_cast<uint4 *>(tile + rnew * ROWB + col) = regs[0];
    *reinterpret_cast<uint4 *>(tile + (rnew + 16u) * ROWB + col) = regs[2];
    *reinterpret_cast<uint4 *>(tile + rold * ROWB + col) = hold[0];
    *reinterpret_cast<uint4 *>(tile + (rold + 16u) * ROWB + col) = hold[1];
    hold[0] = regs[1];
    hold[1] = regs[3];
}

struct Params {
    const uint8_t *data;
    uint64_t       n;
    uint32_t       seg_bytes;
    uint64_t       nsegs;
    const uint64_t *acc;     /* [256] */
    const uint64_t *lut;     /* [8][256] */
    uint64_t       init, self, match;
    uint64_t      *out;
    uint32_t       pad_lds;
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
template <int K>
__device__ inline uint32_t
byte_shl_any(uint32_t v, uint32_t sh)
{
    return byte_shl<(K & 3)>(v, sh);
}

/* W64: 64-bit masks; NLUT exception lookups (bytes 0 .. NLUT-1 of the mask); GRP: accept reads issued GRP bytes ahead */
template <bool W64, int NLUT, int GRP, int MINB>
__global__ __launch_bounds__(256, MINB) void
exp_k(Params P)
{
    typedef typename std::conditional<W64, uint64_t, uint32_t>::type M;
    typedef const __attribute__((address_space(3))) M *lds_m_t;
    constexpr int TILE = 64, WARM = 128;
    __shared__ __attribute__((aligned(16))) M acc_w[256];
    __shared__ __attribute__((aligned(16))) M lut_w[(NLUT ? NLUT : 1) * 256];
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    RowDesc *rows = reinterpret_cast<RowDesc *>(tile + 256 * ROWB);
    const uint32_t tid = threadIdx.x;
    acc_w[tid] = (M) P.acc[tid];
#pragma unroll
    for (int k = 0; k < NLUT; k++) lut_w[k * 256 + tid] = (M) P.lut[k * 256 + tid] | (M) P.init;
    const uint32_t acc_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) acc_w;
    const uint32_t lut_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) lut_w;
    const uint32_t sh = W64 ? 3u : 2u;
    const M init = (M) P.init, match = (M) P.match;
    const uint32_t self_lo = (uint32_t) P.self, self_hi = (uint32_t) (P.self >> 32), src_lo = 0xffffff00u | (uint32_t) P.pad_lds;

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
    uint4    regs[4], hold[2];
    hold[0] = hold[1] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    tile2_fetch(regs, rows, tid, 0);
    for (uint32_t s = 0; s <= nrounds; s++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tile2_store(regs, hold, tile, tid, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (s < nrounds) tile2_fetch(regs, rows, tid, s + 1);
        if (s < lag || s - lag >= nrounds) continue;
        const uint32_t r = s - lag;
        if (!active) continue;
        const int64_t base = seg_a - WARM + (int64_t) r * TILE;
        if (base < 0) continue;
        const uint8_t *srcp = tile + tid * ROWB;
        uint4 piece = make_uint4(0, 0, 0, 0);
        int32_t clean_at = -1;
        M       av[2][GRP];
        auto load_group = [&](int q) {
#pragma unroll
            for (int i = 0; i < GRP; i++) {
                const int      j = q * GRP + i;
                if ((j & 15) == 0) piece = *reinterpret_cast<const uint4 *>(srcp + j);
                const uint32_t word = ((j >> 2) & 3) == 0 ? piece.x : ((j >> 2) & 3) == 1 ? piece.y : ((j >> 2) & 3) == 2 ? piece.z : piece.w;
                const uint32_t a = (j & 3) == 0 ? byte_shl<0>(word, sh) : (j & 3) == 1 ? byte_shl<1>(word, sh)
                                 : (j & 3) == 2 ? byte_shl<2>(word, sh) : byte_shl<3>(word, sh);
                av[q & 1][i] = *(lds_m_t) (uintptr_t) (acc_base + a);
            }
        };
        load_group(0);
#pragma unroll
        for (int q = 0; q < TILE / GRP; q++) {
            if (q + 1 < TILE / GRP) load_group(q + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GRP; i++) {
                const int j = q * GRP + i;
                const M   a = av[q & 1][i];
                if (W64) {
                    uint32_t s_lo = (uint32_t) S, s_hi = (uint32_t) ((uint64_t) S >> 32);
                    const uint32_t t_lo = s_lo & (uint32_t) a, t_hi = s_hi & (uint32_t) ((uint64_t) a >> 32);
                    uint32_t e_lo = (uint32_t) init, e_hi = (uint32_t) ((uint64_t) init >> 32);
                    if (NLUT > 0) {
                        uint64_t e = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) (lut_base + byte_shl<0>(t_lo, sh));
                        e_lo = (uint32_t) e; e_hi = (uint32_t) (e >> 32);
                    }
                    const uint32_t ts_lo = t_lo & src_lo;
                    uint32_t u_lo, u_hi;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(u_lo) : "v"(t_lo), "s"(self_lo), "v"(e_lo));
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(u_hi) : "v"(t_hi), "s"(self_hi), "v"(e_hi));
                    if (NLUT > 1) {
                        uint64_t e = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) (lut_base + 256 * 8 + byte_shl<1>(t_lo, sh));
                        u_lo |= (uint32_t) e; u_hi |= (uint32_t) (e >> 32);
                    }
                    if (NLUT > 2) {
                        uint64_t e = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) (lut_base + 512 * 8 + byte_shl<2>(t_lo, sh));
                        u_lo |= (uint32_t) e; u_hi |= (uint32_t) (e >> 32);
                    }
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(s_lo) : "v"(ts_lo), "v"(u_lo));
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(s_hi) : "v"(t_hi), "v"(u_hi));
                    S = (M) (((uint64_t) s_hi << 32) | s_lo);
                    if ((j & 15) == 15 && (t_lo | t_hi) == 0) clean_at = j + 1;
                } else {
                    const uint32_t t = (uint32_t) S & (uint32_t) a;
                    uint32_t e = (uint32_t) init;
                    if (NLUT > 0) e = *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (lut_base + byte_shl<0>(t, sh));
                    const uint32_t ts = t & src_lo;
                    uint32_t u, sn;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(u) : "v"(t), "s"(self_lo), "v"(e));
                    if (NLUT > 1) u |= *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (lut_base + 256 * 4 + byte_shl<1>(t, sh));
                    if (NLUT > 2) u |= *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (lut_base + 512 * 4 + byte_shl<2>(t, sh));
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(sn) : "v"(ts), "v"(u));
                    S = (M) sn;
                    if ((j & 15) == 15 && t == 0) clean_at = j + 1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (S & match) S &= ~match;
        if (clean_at >= 0) last_clean = base + clean_at;
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
struct Variant { const char *name; kern_t k; uint32_t pad; };

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

#define V(w64, nlut, grp, minb, pad) { #w64 "," #nlut "," #grp "," #minb "," #pad, exp_k<w64, nlut, grp, minb>, pad }
    const Variant vars[] = {
        V(false, 0, 16, 6, 0), V(false, 1, 16, 6, 0), V(false, 2, 16, 6, 0), V(false, 3, 16, 6, 0),
        V(false, 0, 8, 6, 0), V(false, 1, 8, 6, 0), V(false, 2, 8, 6, 0),
        V(true, 0, 8, 6, 0),   V(true, 1, 8, 6, 0),   V(true, 2, 8, 6, 0),   V(true, 3, 8, 6, 0),
        V(true, 0, 4, 6, 0),   V(true, 1, 4, 6, 0),   V(true, 2, 4, 6, 0),
        V(true, 1, 16, 5, 0),  V(true, 2, 16, 5, 0),
        /* occupancy sweep by padding the LDS request */
        V(false, 1, 16, 6, 8192), V(false, 1, 16, 6, 16384), V(false, 1, 16, 6, 28672),
        V(true, 2, 8, 6, 8192), V(true, 2, 8, 6, 16384), V(true, 2, 8, 6, 28672),
        V(false, 0, 16, 6, 16384), V(true, 0, 8, 6, 16384),
    };
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Variant &v : vars) {
        const size_t dyn = 256 * ROWB + 256 * 16 + v.pad;
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, v.k, 256, dyn));
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(v.k)));
        const uint64_t lanes = (uint64_t) ncu * nb * 256 * 63 / 64;
        uint32_t       seg = (uint32_t) ((n + lanes - 1) / lanes);
        seg = (seg + 255) & ~255u;
        Params P;
        memset(&P, 0, sizeof(P));
        P.data = d_data; P.n = n; P.seg_bytes = seg; P.nsegs = n / seg;
        P.acc = d_acc; P.lut = d_lut; P.out = d_out;
        P.init = 0x0000000100000003ull; P.self = 0x1000100010001000ull;
        P.match = 0x8000000080000000ull;
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
        printf("%-22s vgpr %3d lds %6zu wg/cu %d seg %6u grid %6u  avg %.3f ms  best %.3f ms  %.2f TB/s  frac %.3f\n", v.name,
               fa.numRegs, (size_t) fa.sharedSizeBytes + dyn, nb, seg, grid, sum / reps, best, bytes / (sum / reps) * 1e-9,
               bytes / (sum / reps) * 1e-9 / 8.0);
        fflush(stdout);
    }
    return 0;
}
