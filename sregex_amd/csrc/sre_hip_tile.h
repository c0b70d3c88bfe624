/*
 * sre_hip_tile.h — input staging shared by the segment-parallel kernels
 * (sre_hip_scan.hip: table-driven automaton; sre_hip_nfa.hip: bit-parallel NFA).
 *
 * ONE LANE walks ONE SEGMENT, 64 bytes per round; a workgroup of 256 lanes stages
 * its 256 segments through LDS in whole 128-byte lines, one line per row for half
 * a wave per stage (tile_fetch), classified / packed by the fetching lane
 * (tile_store).  Device code only.
 */
#ifndef SRE_HIP_TILE_H
#define SRE_HIP_TILE_H

#include <hip/hip_runtime.h>
#include "sre_hip_scan.h"

/* tile_store<BITS, WIDE>: WIDE = the tile holds 16-bit indices already scaled to a byte
 * offset into a fast-table row (one SDWA add per lookup in the consumer) instead of 8-bit
 * ones; chosen per scanner (sre_scan_tables_t.wide) */
namespace {

/* one row of a workgroup's staging: where the lane's segment (with its warm-up
 * round in front) starts and which byte range of it may be read */
struct __attribute__((aligned(16))) RowDesc {
    uint64_t addr;      /* device address of row offset 0 */
    int32_t  lo;        /* first readable offset */
    int32_t  hi16;      /* last offset at which a whole 16-byte piece may start */
};

typedef uint32_t sre_u32x4 __attribute__((ext_vector_type(4)));
typedef sre_u32x4 __attribute__((aligned(1))) sre_u32x4_unaligned;

/*
 * Staging works on whole 128-byte lines although a lane consumes 64 bytes per
 * round: stage s brings ONE line for each row of half a wave — rows 0..31 of the
 * wave at even stages, rows 32..63 at odd stages — as 4 x 16-byte pieces per
 * lane, 8 adjacent lanes per line.  Every line is requested exactly once and in
 * one piece (two 64-byte requests for the same line one round apart each go to
 * HBM when the L2 has dropped the line in between), and the staging registers
 * stay at 64 bytes per lane.  The upper half-wave therefore runs one round
 * behind the lower one.  Pieces outside a row's valid range read as zeros
 * (their lanes take the exact path there).  The loads are GLOBAL loads
 * (address space 1) on purpose: a flat load also counts on lgkmcnt, so every
 * wait for an LDS lookup in the consumer would wait for the prefetch as well.
 */
__device__ inline void
tile_fetch(uint4 (&regs)[4], const RowDesc *rows, uint32_t tid, uint32_t stage)
{
    /* wave-private staging: the 64 lanes of a wave fetch rows of that same
     * wave, so no workgroup barrier is needed between rounds */
    const uint32_t wbase = (tid & ~63u) + (stage & 1u) * 32u, lane = tid & 63u;
    const int32_t  line_off = (int32_t) ((stage >> 1) * SRE_SCAN_LINE);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t piece = i * 64 + lane;
        const uint32_t row = wbase + piece / 8, col = piece % 8;
        const int32_t  off = line_off + (int32_t) (col * 16);
        const uint4    d = *reinterpret_cast<const uint4 *>(&rows[row]);
        sre_u32x4      v = {0, 0, 0, 0};
        if ((off >= (int32_t) d.z) & (off <= (int32_t) d.w)) {
            const uint64_t a = (((uint64_t) d.y << 32) | d.x) + (uint64_t) (int64_t) off;
            /* rows start wherever the stream does: the load may be unaligned,
             * which the hardware handles */
            v = *reinterpret_cast<const __attribute__((address_space(1))) sre_u32x4_unaligned *>(a);
        }
        regs[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

/* Stage one line per row of half a wave into LDS.  The raw bytes are classified
 * and packed HERE, by the lane that fetched them (16 independent class lookups
 * per piece), so the tile holds ready-made fast-table indices and the consuming
 * lane's dependent chain is table lookups only.  A tile row holds the two
 * 64-byte halves of the row's current line back to back plus a 16-byte pad,
 * which makes the consumer's 16-byte reads bank-conflict free. */
template <int BITS, bool WIDE>
__device__ inline void
tile_store(const uint4 (&regs)[4], uint8_t *tile, const uint16_t (*clsx)[256], uint32_t tid, uint32_t stage)
{
    constexpr int      STRIDE = 8 / BITS;                       /* input bytes per index */
    constexpr uint32_t HALFB = SRE_SCAN_ROUND / STRIDE * (WIDE ? 2 : 1);
    constexpr uint32_t ROWB = 2 * HALFB + 16;
    constexpr uint32_t PIECEB = HALFB / 4;                      /* index bytes per 16 input bytes */
    const uint32_t wbase = (tid & ~63u) + (stage & 1u) * 32u, lane = tid & 63u;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t piece = i * 64 + lane;
        const uint32_t row = wbase + piece / 8, col = piece % 8;
        const uint32_t words[4] = {regs[i].x, regs[i].y, regs[i].z, regs[i].w};
        uint8_t       *dst = tile + row * ROWB + col * PIECEB;
        if (BITS == 8) {
            *reinterpret_cast<uint4 *>(dst) = regs[i];
        } else {
            /* one index per STRIDE input bytes: table u holds the class shifted
             * to its place, so an index is the OR of its lookups.  With few
             * classes the index is stored as a 16-bit word already scaled to a
             * byte offset into a fast-table row, otherwise as a byte. */
            constexpr int NIDX = 16 / STRIDE;                   /* indices per piece */
            constexpr int PERW = WIDE ? 2 : 4;                  /* indices per 32-bit word */
            uint32_t      out[NIDX / PERW];
#pragma unroll
            for (int k = 0; k < NIDX; k++) {
                uint32_t v = 0;
#pragma unroll
                for (int u = 0; u < STRIDE; u++) {
                    const int      b = k * STRIDE + u;
                    const uint32_t c = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                    v |= (uint32_t) clsx[u][c];
                }
                if (k % PERW) out[k / PERW] |= v << ((k % PERW) * (32 / PERW)); else out[k / PERW] = v;
            }
            constexpr int NW = NIDX / PERW;
            if (PIECEB == 4) *reinterpret_cast<uint32_t *>(dst) = out[0];
            else if (PIECEB == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(out[0], out[1 % NW]);
            else *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1 % NW], out[2 % NW], out[3 % NW]);
        }
    }
}

/*
 * HALF-LINE staging (round 3; the shift-and NFA kernel): the tile keeps 64 bytes per row instead
 * of a whole line, so a workgroup's tile is 20 KiB instead of 36 and twice as many workgroups
 * share a CU — these kernels are bound by the latency of one lane's dependent chain, and waves
 * per SIMD are what hides it.  Lines are still requested whole and once: stage s brings the line
 * (s >> 1) of the rows of half (s & 1) of the wave; lanes 4q .. 4q + 3 read 64 contiguous bytes,
 * pieces 0 / 2 the FIRST halves of the lines of rows q and q + 16, pieces 1 / 3 their SECOND
 * halves.  tile2_store puts the first halves into LDS now and keeps the second halves in
 * registers (`hold`) for one stage: every row of the wave gets 64 fresh bytes per stage, the rows
 * of half (s & 1) the first half of a new line, the others the second half of theirs.  The upper
 * half-wave therefore runs one round behind the lower one, as with tile_fetch.
 */
constexpr uint32_t SRE_TILE2_ROWB = SRE_SCAN_ROUND + 16;    /* 64 bytes + pad: 16-byte reads of a row are conflict-free */

__device__ inline void
tile2_fetch(uint4 (&regs)[4], const RowDesc *rows, uint32_t tid, uint32_t stage)
{
    const uint32_t wbase = (tid & ~63u) + (stage & 1u) * 32u, lane = tid & 63u;
    const int32_t  line_off = (int32_t) ((stage >> 1) * SRE_SCAN_LINE);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t row = wbase + (lane >> 2) + 16u * (i >> 1), col = (lane & 3u) + 4u * (i & 1u);
        const int32_t  off = line_off + (int32_t) (col * 16);
        const uint4    d = *reinterpret_cast<const uint4 *>(&rows[row]);
        sre_u32x4      v = {0, 0, 0, 0};
        if ((off >= (int32_t) d.z) & (off <= (int32_t) d.w)) {
            const uint64_t a = (((uint64_t) d.y << 32) | d.x) + (uint64_t) (int64_t) off;
            v = *reinterpret_cast<const __attribute__((address_space(1))) sre_u32x4_unaligned *>(a);
        }
        regs[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

__device__ inline void
tile2_store(const uint4 (&regs)[4], uint4 (&hold)[2], uint8_t *tile, uint32_t tid, uint32_t stage)
{
    const uint32_t wave = tid & ~63u, lane = tid & 63u, h = stage & 1u;
    const uint32_t rnew = wave + h * 32u + (lane >> 2), rold = wave + (1u - h) * 32u + (lane >> 2);
    const uint32_t col = (lane & 3u) * 16u;
    *reinterpret_cast<uint4 *>(tile + rnew * SRE_TILE2_ROWB + col) = regs[0];
    *reinterpret_cast<uint4 *>(tile + (rnew + 16u) * SRE_TILE2_ROWB + col) = regs[2];
    *reinterpret_cast<uint4 *>(tile + rold * SRE_TILE2_ROWB + col) = hold[0];
    *reinterpret_cast<uint4 *>(tile + (rold + 16u) * SRE_TILE2_ROWB + col) = hold[1];
    hold[0] = regs[1];
    hold[1] = regs[3];
}

}  // namespace

#endif
