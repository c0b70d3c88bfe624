/*
 * sre_hip_nfa.hip — the bit-parallel NFA scanner (gfx950): the throughput tier
 * for programs whose ordered-list automaton is too large for the table-driven
 * scanner (sre_hip_scan.hip) but whose list-able threads fit a 64-bit mask.
 *
 * Same data layout as the table-driven scanner: a stream is cut into segments,
 * ONE LANE walks ONE SEGMENT, 64 bytes per round, staged through LDS in whole
 * 128-byte lines (sre_hip_tile.h).  The lane's state is the SET of live threads
 * as a bit mask (sre_nfa.h); per input byte
 *      T = S & accept[byte];   S = OR_k follow[k][(T >> 8k) & 255]
 * with both tables in LDS: NSLICE + 1 lookups and no classification pass (the
 * accept table is indexed by the raw byte).  MATCH bits are sticky in the device
 * tables (accepted by every byte, following to themselves), so the common round
 * tests for an event once, after its 64 steps; a round with an event is replayed
 * byte by byte from its entry set to find the exact step.
 *
 * Exactness: sets are exact for Thompson (sre_vm_thompson.c:88-258) and, for
 * Pike, up to the first MATCH event (sre_vm_pike.c:535-553 cuts threads only
 * then).  The lane reports the first event of its segment and a CLEAN position
 * in front of it — one where only the ".*?" thread consumed the previous byte,
 * so the list there is the freshly seeded initial closure and nothing else; the
 * exact VM (sre_hip_vm.hip, sre_k_pike_window) then runs from that position.
 * Entry sets are speculative (a 128-byte warm-up, which can only UNDER-estimate
 * the true set: the step is monotone) and verified by the chain check below;
 * nothing is reported from an unverified segment.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include "sre_hip_nfa.h"
#include "sre_hip_tile.h"

#define RC_DECLINED (-5)
#define RC_ERROR    (-1)

namespace {

__device__ inline uint32_t
nfa_stream_of(const sre_scan_geom_t &G, uint64_t g)
{
    uint32_t a = 0, b = G.nstreams;
    while (b - a > 1) {
        uint32_t m = (a + b) >> 1;
        if (geom_first(G, m) <= g) a = m; else b = m;
    }
    return a;
}

/* ((v >> 8K) & 255) << sh in ONE instruction: the SDWA byte select feeds the
 * shifter, so a table lookup by a byte of a mask costs one address op */
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

/* OR of the follow-table entries selected by the bytes of t: slice K of the table
 * at byte offset K * 256 * sizeof(M) behind `fol_base` (an LDS address) */
template <typename M, int NSLICE, int K>
struct FollowOr {
    typedef const __attribute__((address_space(3))) M *lds_m_t;
    static __device__ inline M get(M t, uint32_t fol_base, uint32_t sh)
    {
        const uint32_t word = K < 4 ? (uint32_t) t : (uint32_t) ((uint64_t) t >> 32);
        const M        v = *(lds_m_t) (uintptr_t) (fol_base + K * 256 * (uint32_t) sizeof(M) + byte_shl<(K & 3)>(word, sh));
        return v | FollowOr<M, NSLICE, K + 1>::get(t, fol_base, sh);
    }
};
template <typename M, int NSLICE>
struct FollowOr<M, NSLICE, NSLICE> {
    static __device__ inline M get(M, uint32_t, uint32_t) { return 0; }
};

/* an entry of the accept table: the mask of threads that consume the byte and, with
 * look-ahead assertions in the program, what the byte means to them */
template <typename M, bool LA> struct AccEntry { M a; };
template <> struct __attribute__((aligned(8))) AccEntry<uint32_t, true> {
    uint32_t a;
    uint32_t kk;        /* [15] the byte can start a match  [14:0] byte offset of its column (kind as the
                           byte AT a position) in an expansion table row  [31:16] byte offset of its
                           row (kind as the byte IN FRONT of the next position) */
};
template <> struct __attribute__((aligned(16))) AccEntry<uint64_t, true> {
    uint64_t a;
    uint32_t kk, pad;
};

/*
 * MODE 0: Thompson (events only); MODE 1: Pike first match (events + clean
 * positions).  NSLICE = byte slices of the mask; masks are 32-bit up to 4
 * slices, 64-bit above.  LA: the program has look-ahead assertions; their bits are
 * byte NSLICE - 1 of the mask (sre_nfa.h).
 */
template <int MODE, int NSLICE, bool LA>
__global__ __launch_bounds__(SRE_SCAN_BLOCK) void
sre_k_nfa(sre_nfa_tables_t T, sre_scan_geom_t G, sre_nfa_summary_t *__restrict__ sum,
          const int64_t *__restrict__ lo, const uint64_t *__restrict__ belief,
          const uint8_t *__restrict__ bvalid)
{
    typedef typename std::conditional<(NSLICE <= 4), uint32_t, uint64_t>::type M;
    typedef AccEntry<M, LA> Acc;
    typedef const __attribute__((address_space(3))) M *lds_m_t;
    constexpr int      TILE = SRE_SCAN_ROUND;
    constexpr int      WARM = SRE_SCAN_LINE;
    constexpr uint32_t ROWRAW = TILE;               /* raw bytes per round */
    constexpr uint32_t ROWB = 2 * ROWRAW + 16;      /* see tile_store */
    constexpr int      FSL = LA ? NSLICE - 1 : NSLICE;      /* slices that can hold consuming threads */
    constexpr uint32_t XROW = 4 * 256 * (uint32_t) sizeof(M), XCOL = 256 * (uint32_t) sizeof(M);
    /* static LDS: table addresses are compile-time constants and fold into the
     * offset field of the lookups; the tile (and the row descriptors behind it) are
     * dynamic: with the expansion tables the static part alone nears 64 KiB */
    __shared__ __attribute__((aligned(16))) Acc acc_w[256];
    __shared__ __attribute__((aligned(16))) M fol_w[(FSL ? FSL : 1) * 256];
    __shared__ __attribute__((aligned(16))) M exp_w[LA ? 16 * 256 : 1];
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    RowDesc *rows = reinterpret_cast<RowDesc *>(tile + SRE_SCAN_BLOCK * ROWB);

    const uint32_t tid = threadIdx.x;
    const M  match = (M) T.match_bits, any = (M) T.any_bits;
    /* sticky MATCH bits: see the file comment */
    acc_w[tid].a = (M) T.accept[tid] | match;
    if (LA) {
        const uint32_t kd = T.kind[tid];
        reinterpret_cast<AccEntry<M, true> *>(acc_w)[tid].kk =
            ((kd & 3u) * XCOL) | ((kd & 4u) ? 0x8000u : 0u) | (((kd & 3u) * XROW) << 16);
        for (uint32_t i = tid; i < 16 * 256; i += SRE_SCAN_BLOCK) exp_w[i] = (M) T.expand[i];
    }
#pragma unroll
    for (int k = 0; k < FSL; k++) {
        const M slice_match = (M) ((T.match_bits >> (8 * k)) & 0xffu) & (M) tid;
        fol_w[k * 256 + tid] = (M) T.follow[k * 256 + tid] | (slice_match << (8 * k));
    }
    const uint32_t acc_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) Acc *) acc_w;
    const uint32_t fol_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) fol_w;
    const uint32_t exp_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) M *) exp_w;
    const uint32_t sh = sizeof(M) == 4 ? 2u : 3u;      /* log2 of a table entry, in a register for SDWA */
    const uint32_t sha = sizeof(Acc) == 4 ? 2u : sizeof(Acc) == 8 ? 3u : 4u;

    /* ---- which segment am I ---- */
    const uint64_t g = (uint64_t) blockIdx.x * SRE_SCAN_BLOCK + tid;
    bool           active = g < G.nsegs;
    uint32_t       sidx = 0;
    uint64_t       k = 0;
    if (active) {
        sidx = nfa_stream_of(G, g);
        k = g - geom_first(G, sidx);
        if (lo != nullptr && (lo[sidx] < 0 || (int64_t) k < lo[sidx])) active = false;
    }

    const uint8_t *data = nullptr;
    int64_t        n = 0, seg_a = 0, seg_b = 0;
    M              S = 0, s_in = 0;
    bool           warm = false, finished = false, last_seg = false;
    int64_t        first_ev = -1, last_clean = -1;
    int32_t        clean_mode = 0;
    uint32_t       prev_off = 3u * XROW;        /* LA: expansion-table row of the byte in front (3: stream start) */
    RowDesc        mine;
    mine.addr = 0;
    mine.lo = 0;
    mine.hi16 = -1;
    /* per-stream start conditions (find-all rounds), else the batch's */
    const uint32_t sfl = (active && G.sflags != nullptr) ? G.sflags[sidx] : 0u;
    const uint32_t v_init = G.sflags != nullptr ? SRE_SFLAG_INIT(sfl) : G.init_variant;
    const uint32_t v_snap = G.sflags != nullptr ? SRE_SFLAG_SNAP(sfl) : G.init_variant;
    const bool     no_eof = G.sflags != nullptr ? (sfl & SRE_SFLAG_NO_EOF) != 0 : (G.flags & SRE_GEOM_NO_EOF) != 0;
    if (active) {
        data = geom_ptr(G, sidx);
        n = (int64_t) geom_len(G, sidx);
        seg_a = (int64_t) k * G.seg_bytes;
        seg_b = seg_a + G.seg_bytes;
        last_seg = (k + 1 == geom_first(G, sidx + 1) - geom_first(G, sidx));
        if (seg_b > n) seg_b = n;
        if (k == 0) {
            S = (M) T.init[v_init];
            last_clean = 0;                     /* the search starts here */
            clean_mode = (int32_t) SRE_SFLAG_MODE(sfl);
        } else if (lo != nullptr && ((int64_t) k == lo[sidx] || bvalid[g])) {
            /* exact carry of the verified prefix, or (later segments of a fix-up
             * round) what the previous round's lane in front of this one ended in */
            S = (M) belief[g];
        } else {
            warm = true;
            /* a search that is (re)started in the middle of a stream: ^ false */
            S = (M) T.init[seg_a <= WARM ? v_init : 2];
        }
        s_in = S;
        mine.addr = (uint64_t) reinterpret_cast<uintptr_t>(data) + (uint64_t) (seg_a - WARM);
        mine.lo = warm ? (seg_a >= WARM ? 0 : (int32_t) (WARM - seg_a)) : WARM;
        mine.hi16 = (int32_t) (WARM + (seg_b - seg_a)) - 16;
        if (LA) {
            /* the kind of the byte in front of the first byte this lane steps over */
            const int64_t first_pos = warm ? (seg_a >= WARM ? seg_a - WARM : 0) : seg_a;
            if (first_pos > 0) prev_off = (T.kind[data[first_pos - 1]] & 3u) * XROW;
        }
    }
    rows[tid] = mine;

    /*
     * How the reference ARRIVES at a clean position q (the list there is the fresh
     * initial closure): by an ordinary step, or as the target of its leading-byte
     * skip (sre_vm_pike.c:256-309).  The two differ in one flag: a skip target is
     * stepped without a new "is this the initial state" check and keeps
     * seen_start_state set (:286-306 fall through to :312), which decides whether a
     * later check can re-seed a search that already holds a match.  The skip fires
     * at q - 1 iff the list there equals the snapshot taken at offset 0 (:266-273)
     * and the byte at q - 1 cannot start a match (without look-ahead assertions it
     * never can, or q would not be clean; with them a thread that could consume it
     * may have been held back by its assertion):
     *   the byte at q - 1 can start a match                     -> an ordinary step (0)
     *   the SET in front of q - 1 differs from the snapshot's   -> an ordinary step (0)
     *   it is equal and q - 1 is clean too (same list, fresh)   -> a skip target   (1)
     *   equal as a set but not known to be the same list        -> unusable        (-1)
     */
    const M snap = (M) T.init[v_snap];
    auto clean_kind = [&](M s_before, bool prev_clean, bool leading) -> int {
        if (leading || s_before != snap) return 0;
        return prev_clean ? 1 : -1;
    };
    /* the accept-table entry of the byte selected from a word of the row: j = its byte index */
    struct AccVal { M a; uint32_t kk; };
    auto accept_of = [&](uint32_t word, int j) -> AccVal {
        const uint32_t a = (j & 3) == 0 ? byte_shl<0>(word, sha) : (j & 3) == 1 ? byte_shl<1>(word, sha)
                         : (j & 3) == 2 ? byte_shl<2>(word, sha) : byte_shl<3>(word, sha);
        AccVal v;
        v.a = *(lds_m_t) (uintptr_t) (acc_base + a);
        v.kk = LA ? *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (acc_base + a + (uint32_t) sizeof(M))
                  : 0u;
        return v;
    };
    /* LA: the assertions of the list that hold between the byte in front and this one put
     * their continuations into the list (one lookup: sre_nfa.h) */
    auto expand = [&](M s, uint32_t col_off) -> M {
        const uint32_t word = (NSLICE - 1) < 4 ? (uint32_t) s : (uint32_t) ((uint64_t) s >> 32);
        return s | *(lds_m_t) (uintptr_t) (exp_base + prev_off + col_off + byte_shl<((NSLICE - 1) & 3)>(word, sh));
    };
    auto step = [&](M s, const AccVal &e, M &t_out) -> M {
        if (LA) {
            s = expand(s, e.kk & 0x7fffu);
            prev_off = e.kk >> 16;
        }
        const M t = s & e.a;
        t_out = t;
        return FollowOr<M, FSL, 0>::get(t, fol_base, sh);
    };
    auto is_leading = [&](const AccVal &e) -> bool { return LA && (e.kk & 0x8000u) != 0; };

    const uint32_t nrounds = WARM / TILE + G.seg_bytes / TILE;
    const uint32_t lag = (tid >> 5) & 1u;
    uint4          regs[4];
    __syncthreads();                        /* tables and row descriptors are complete */
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
        const bool     warm_round = (r < WARM / TILE);
        if (!active || finished || (warm_round && !warm)) continue;
        const int64_t base = seg_a - WARM + (int64_t) r * TILE;
        if (base >= seg_b || base < 0) continue;

        uint32_t roww[TILE / 4];
        {
            const uint8_t *src = tile + tid * ROWB + (r & 1u) * ROWRAW;
#pragma unroll
            for (int x = 0; x < TILE / 16; x++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + 16 * x);
                roww[4 * x] = v.x; roww[4 * x + 1] = v.y; roww[4 * x + 2] = v.z; roww[4 * x + 3] = v.w;
            }
        }
        const M        s0 = S;
        const uint32_t prev_off0 = prev_off;
        if (base + TILE <= seg_b) {
            /* the common round: 64 steps, then one look at the sticky MATCH bits */
            int32_t clean_at = -1, clean_how = 0;
            M       t14 = 0;
#pragma unroll
            for (int j = 0; j < TILE; j++) {
                M         t;
                const M   s_before = S;
                const AccVal e = accept_of(roww[j >> 2], j);
                S = step(S, e, t);
                if (MODE == 1 && (j & 15) == 14) t14 = t;
                /* clean positions are sampled at the end of every 16-byte group */
                if (MODE == 1 && (j & 15) == 15 && t <= any) {
                    const int how = clean_kind(s_before, t14 <= any, is_leading(e));
                    if (how >= 0) {
                        clean_at = j + 1;
                        clean_how = how;
                    }
                }
            }
            if (!(S & match)) {
                if (warm_round) {
                    if (r + 1 == WARM / TILE) s_in = S;
                } else if (MODE == 1 && clean_at >= 0) {
                    last_clean = base + clean_at;
                    clean_mode = clean_how;
                }
                continue;
            }
            if (warm_round) {
                /* an event in front of the segment is somebody else's; drop the
                 * sticky bits and go on */
                S &= ~match;
                if (r + 1 == WARM / TILE) s_in = S;
                continue;
            }
            S = s0;
            prev_off = prev_off0;
        }
        /* byte by byte: a round with an event, or the ragged end of the stream */
        {
            const int64_t end = base + TILE <= seg_b ? base + TILE : seg_b;
            bool          prev_clean = (base == 0);     /* the list at offset 0 is the initial one */
#pragma unroll 1
            for (int64_t p = base; p < end; p++) {
                M              t;
                const M        s_before = S;
                /* from memory, not from the tile: a 16-byte piece that crosses the end
                 * of the stream is not staged (sre_hip_tile.h) */
                const uint32_t b = data[p];
                const AccVal   e = accept_of(b, 0);
                S = step(S, e, t);
                if (S & match) {
                    if (!warm_round) {
                        first_ev = p;
                        finished = true;
                        break;
                    }
                    S &= ~match;
                    prev_clean = false;
                } else if (MODE == 1 && !warm_round && t <= any) {
                    const int how = clean_kind(s_before, prev_clean, is_leading(e));
                    if (how >= 0) {
                        last_clean = p + 1;
                        clean_mode = how;
                    }
                    prev_clean = true;
                } else {
                    prev_clean = false;
                }
            }
            if (warm_round && r + 1 == WARM / TILE) s_in = S;
        }
    }

    if (!active) return;
    if (LA && last_seg && !finished && !no_eof) {
        /* the extra iteration at end of input (sre_vm_pike.c:235): assertions that hold
         * in front of the end list their continuations; a MATCH among them is an event */
        if (expand(S, 3u * XCOL) & match) first_ev = n;
    }
    sre_nfa_summary_t out;
    out.s_in = (uint64_t) s_in;
    out.s_out = (uint64_t) S;
    out.first_ev = first_ev;
    out.last_clean = last_clean < 0 ? -1 : last_clean * 2 + clean_mode;
    sum[g] = out;
}

/* ===================================================================== shift-and kernel */

/*
 * sre_k_nfa_sa<W64, CARRY, MASKED, EVACC, NLUT> — the SHIFT-AND form of the set step (sre_nfa.h):
 *
 *      t  = S & accept[byte]
 *      S' = ((t [& shift_src]) << 1) | (t & self) | seed | OR_k lut[k][hot byte k of t]
 *
 * One accept read per input byte that does not depend on the state (issued a group of bytes
 * ahead), three ALU operations per 32 bits of mask, and NLUT (0..3) dependent lookups where the
 * plain slices take nbits / 8.  The hot bytes are gathered by one v_perm_b32 (selector in a kernel
 * argument), so any layout the host builder finds runs on the same code.  Staging: half a line per
 * row (sre_hip_tile.h tile2_*), 20 KiB per workgroup, five or six workgroups per CU.
 * Both modes in one kernel: clean positions are sampled every 16 bytes, which is noise.
 * Summaries, beliefs and the chain check are those of sre_k_nfa.
 */
template <bool W64, bool CARRY, bool MASKED, bool EVACC, int NLUT, bool LA>
__global__ __launch_bounds__(SRE_SCAN_BLOCK, (W64 || LA ? 5 : 6)) void
sre_k_nfa_sa(sre_nfa_sa_tables_t T, sre_scan_geom_t G, sre_nfa_summary_t *__restrict__ sum,
             const int64_t *__restrict__ lo, const uint64_t *__restrict__ belief,
             const uint8_t *__restrict__ bvalid)
{
    typedef typename std::conditional<W64, uint64_t, uint32_t>::type E; /* a table entry */
    constexpr int      TILE = SRE_SCAN_ROUND;
    constexpr int      WARM = SRE_SCAN_LINE;
    constexpr uint32_t ROWB = SRE_TILE2_ROWB;
    constexpr int      GRP = LA ? (W64 ? 2 : 4) : (W64 ? 4 : 8);    /* accept reads in flight ahead of the chain */
    constexpr uint32_t ESZ = (uint32_t) sizeof(E);
    /* LA: an accept entry also says what the byte means to the look-ahead assertions (kk, as in sre_k_nfa) */
    constexpr uint32_t ASZ = LA ? 2 * ESZ : ESZ;
    static_assert(!LA || (MASKED && !EVACC), "look-ahead forms are masked and keep MATCH bits");
    __shared__ __attribute__((aligned(16))) uint8_t acc_raw[256 * ASZ];
    __shared__ __attribute__((aligned(16))) E lut_w[(NLUT ? NLUT : 1) * 256];
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    RowDesc *rows = reinterpret_cast<RowDesc *>(tile + SRE_SCAN_BLOCK * ROWB);
    /* LA: the expansion table [16][1 << nassert] behind the row descriptors */
    E       *exp_w = reinterpret_cast<E *>(tile + SRE_SCAN_BLOCK * ROWB + SRE_SCAN_BLOCK * 16);
    const uint32_t XCOL = LA ? (ESZ << T.nassert) : 0u, XROW = 4u * XCOL;

    const uint32_t tid = threadIdx.x;
    auto lo32 = [](uint64_t v) { return (uint32_t) v; };
    auto hi32 = [](uint64_t v) { return (uint32_t) (v >> 32); };
    auto entry = [&](uint64_t v) -> E { return (E) v; };
    /* sticky MATCH bits accept every byte and list themselves in the host tables already */
    *reinterpret_cast<E *>(acc_raw + tid * ASZ) = entry(T.accept[tid]);
    if (LA) {
        const uint32_t kd = T.kind[tid];
        *reinterpret_cast<uint32_t *>(acc_raw + tid * ASZ + ESZ) =
            ((kd & 3u) * XCOL) | ((kd & 4u) ? 0x8000u : 0u) | (((kd & 3u) * XROW) << 16);
        for (uint32_t i = tid; i < (16u << T.nassert); i += SRE_SCAN_BLOCK) exp_w[i] = entry(T.expand[i]);
    }
#pragma unroll
    for (int k = 0; k < NLUT; k++) lut_w[k * 256 + tid] = entry(T.lut[k * 256 + tid] | (k == 0 ? T.seed : 0));
    const uint32_t acc_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) acc_raw;
    const uint32_t lut_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) E *) lut_w;
    const uint32_t exp_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) E *) exp_w;
    const uint32_t sh = W64 ? 3u : 2u;          /* log2 of a table entry, in a register for SDWA */
    const uint32_t sha = LA ? sh + 1u : sh;     /* ... of an accept entry */
    const uint32_t amask = LA ? ((1u << T.nassert) - 1u) : 0u;
    uint32_t       prev_off = 3u * XROW;        /* LA: the expansion-table row of the byte in front (3: stream start) */
    const uint32_t self_lo = lo32(T.self), self_hi = hi32(T.self), src_lo = lo32(T.shift_src), src_hi = hi32(T.shift_src);
    const uint32_t seed_lo = NLUT ? 0u : lo32(T.seed), seed_hi = NLUT ? 0u : hi32(T.seed);
    const uint32_t nany_lo = ~lo32(T.any_bits), nany_hi = ~hi32(T.any_bits);
    const uint32_t ev_lo = EVACC ? lo32(T.msrc) : lo32(T.match_bits), ev_hi = EVACC ? hi32(T.msrc) : hi32(T.match_bits);
    const uint32_t val_lo = lo32(T.valid), val_hi = W64 ? hi32(T.valid) : 0u;
    const uint32_t perm = T.perm;

    /* ---- which segment am I ---- */
    const uint64_t g = (uint64_t) blockIdx.x * SRE_SCAN_BLOCK + tid;
    bool           active = g < G.nsegs;
    uint32_t       sidx = 0;
    uint64_t       k = 0;
    if (active) {
        sidx = nfa_stream_of(G, g);
        k = g - geom_first(G, sidx);
        if (lo != nullptr && (lo[sidx] < 0 || (int64_t) k < lo[sidx])) active = false;
    }

    const uint8_t *data = nullptr;
    int64_t        n = 0, seg_a = 0, seg_b = 0;
    uint32_t       s_lo = 0, s_hi = 0;
    uint64_t       s_in = 0;
    bool           warm = false, finished = false;
    int64_t        first_ev = -1, last_clean = -1;
    int32_t        clean_mode = 0;
    RowDesc        mine;
    mine.addr = 0;
    mine.lo = 0;
    mine.hi16 = -1;
    /* per-stream start conditions (find-all rounds), else the batch's */
    const uint32_t sfl = (active && G.sflags != nullptr) ? G.sflags[sidx] : 0u;
    const uint32_t v_init = G.sflags != nullptr ? SRE_SFLAG_INIT(sfl) : G.init_variant;
    const uint32_t v_snap = G.sflags != nullptr ? SRE_SFLAG_SNAP(sfl) : G.init_variant;
    const bool     no_eof = G.sflags != nullptr ? (sfl & SRE_SFLAG_NO_EOF) != 0 : (G.flags & SRE_GEOM_NO_EOF) != 0;
    bool           last_seg = false;
    if (active) {
        data = geom_ptr(G, sidx);
        n = (int64_t) geom_len(G, sidx);
        seg_a = (int64_t) k * G.seg_bytes;
        seg_b = seg_a + G.seg_bytes;
        last_seg = (k + 1 == geom_first(G, sidx + 1) - geom_first(G, sidx));
        if (seg_b > n) seg_b = n;
        uint64_t S;
        if (k == 0) {
            S = T.init[v_init];
            last_clean = 0;                     /* the search starts here */
            clean_mode = (int32_t) SRE_SFLAG_MODE(sfl);
        } else if (lo != nullptr && ((int64_t) k == lo[sidx] || bvalid[g])) {
            S = belief[g];
        } else {
            warm = true;
            S = T.init[seg_a <= WARM ? v_init : 2];
        }
        s_in = S;
        s_lo = lo32(S);
        s_hi = hi32(S);
        mine.addr = (uint64_t) reinterpret_cast<uintptr_t>(data) + (uint64_t) (seg_a - WARM);
        mine.lo = warm ? (seg_a >= WARM ? 0 : (int32_t) (WARM - seg_a)) : WARM;
        mine.hi16 = (int32_t) (WARM + (seg_b - seg_a)) - 16;
        if (LA) {
            /* the kind of the byte in front of the first byte this lane steps over */
            const int64_t first_pos = warm ? (seg_a >= WARM ? seg_a - WARM : 0) : seg_a;
            if (first_pos > 0) prev_off = (T.kind[data[first_pos - 1]] & 3u) * XROW;
        }
    }
    rows[tid] = mine;

    const uint64_t snap = T.init[v_snap];
    const uint32_t snap_lo = lo32(snap), snap_hi = hi32(snap);
    /* how the reference arrives at a clean position: see sre_k_nfa (no look-ahead assertions here, so
     * the byte in front of a clean position never is a leading byte) */
    auto clean_kind = [&](bool before_is_snap, bool prev_clean, bool leading) -> int {
        if (leading || !before_is_snap) return 0;       /* (leading: only with look-ahead assertions, see sre_k_nfa) */
        return prev_clean ? 1 : -1;
    };
    uint32_t evv = 0;           /* EVACC: threads that reached MATCH in this round */
    /* one step; a = the byte's accept entry; returns whether only the ".*?" thread consumed the byte */
    /* LA: the assertions of the list that hold between the byte in front and this one list their
     * continuations (one lookup by the assertion bits, bits 0 .. of the mask) */
    auto expand = [&](uint32_t col_off) {
        const uint32_t at = exp_base + prev_off + col_off + ((s_lo & amask) << sh);
        if constexpr (W64) {
            const uint64_t v = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) at;
            s_lo |= (uint32_t) v;
            s_hi |= (uint32_t) (v >> 32);
        } else {
            s_lo |= *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) at;
        }
    };
    auto step = [&](uint32_t a_lo, uint32_t a_hi, uint32_t kk) -> bool {
        if (LA) {
            /* (a wave in which no lane lists an assertion skips the lookup and its round trip: behind a
             * trailing `$` or `\b` that is nearly every byte) */
            if (__builtin_amdgcn_ballot_w64((s_lo & amask) != 0) != 0) expand(kk & 0x7fffu);
            prev_off = kk >> 16;
        }
        const uint32_t t_lo = s_lo & a_lo, t_hi = W64 ? (s_hi & a_hi) : 0u;
        uint32_t       e_lo = seed_lo, e_hi = seed_hi;
        if (NLUT > 0) {
            const uint32_t hot = __builtin_amdgcn_perm(W64 ? t_hi : t_lo, t_lo, perm);
#pragma unroll
            for (int q = 0; q < NLUT; q++) {
                const uint32_t off = q == 0 ? byte_shl<0>(hot, sh) : q == 1 ? byte_shl<1>(hot, sh) : byte_shl<2>(hot, sh);
                const uint32_t at = lut_base + (uint32_t) q * 256u * ESZ + off;
                if constexpr (W64) {
                    const uint64_t v = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) at;
                    e_lo |= (uint32_t) v;
                    e_hi |= (uint32_t) (v >> 32);
                } else {
                    e_lo |= *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) at;
                }
            }
        }
        const uint32_t ts_lo = MASKED ? (t_lo & src_lo) : t_lo, ts_hi = MASKED ? (t_hi & src_hi) : t_hi;
        const uint32_t u_lo = (t_lo & self_lo) | e_lo;
        s_lo = (ts_lo << 1) | u_lo;
        if (W64) {
            const uint32_t u_hi = (t_hi & self_hi) | e_hi;
            s_hi = (CARRY ? __builtin_amdgcn_alignbit(ts_hi, ts_lo, 31) : (ts_hi << 1)) | u_hi;
        }
        if (EVACC) {
            /* one v_and_or_b32 per word, kept in order: left to itself the compiler turns the 64
             * accumulations of a round into a tree and spills the operands */
            asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(evv) : "v"(t_lo), "v"(ev_lo));
            if (W64) asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(evv) : "v"(t_hi), "v"(ev_hi));
        }
        return ((t_lo & nany_lo) | (W64 ? (t_hi & nany_hi) : 0u)) == 0;
    };
    auto event = [&]() -> bool { return EVACC ? evv != 0 : ((s_lo & ev_lo) | (W64 ? (s_hi & ev_hi) : 0u)) != 0; };
    auto drop_event = [&]() {
        if (EVACC) evv = 0;
        else {
            s_lo &= ~ev_lo;
            s_hi &= ~ev_hi;
        }
    };
    auto is_snap = [&](uint32_t b_lo, uint32_t b_hi) -> bool {
        return (((b_lo ^ snap_lo) & val_lo) | (W64 ? ((b_hi ^ snap_hi) & val_hi) : 0u)) == 0;
    };
    auto state64 = [&]() -> uint64_t { return ((uint64_t) (s_hi & val_hi) << 32) | (s_lo & val_lo); };
    auto accept_at = [&](uint32_t byte_addr, uint32_t &a_lo, uint32_t &a_hi, uint32_t &kk) {
        kk = 0;
        if constexpr (W64) {
            if constexpr (LA) {
                const sre_u32x4 v = *(const __attribute__((address_space(3))) sre_u32x4 *) (uintptr_t) (acc_base + byte_addr);
                a_lo = v.x;
                a_hi = v.y;
                kk = v.z;
            } else {
                const uint64_t v = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) (acc_base + byte_addr);
                a_lo = (uint32_t) v;
                a_hi = (uint32_t) (v >> 32);
            }
        } else {
            if constexpr (LA) {
                const uint64_t v = *(const __attribute__((address_space(3))) uint64_t *) (uintptr_t) (acc_base + byte_addr);
                a_lo = (uint32_t) v;
                kk = (uint32_t) (v >> 32);
            } else {
                a_lo = *(const __attribute__((address_space(3))) uint32_t *) (uintptr_t) (acc_base + byte_addr);
            }
            a_hi = 0;
        }
    };

    const uint32_t nrounds = WARM / TILE + G.seg_bytes / TILE;
    const uint32_t lag = (tid >> 5) & 1u;
    uint4          regs[4], hold[2];
    hold[0] = hold[1] = make_uint4(0, 0, 0, 0);
    __syncthreads();                        /* tables and row descriptors are complete */
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
        const bool     warm_round = (r < WARM / TILE);
        if (!active || finished || (warm_round && !warm)) continue;
        const int64_t base = seg_a - WARM + (int64_t) r * TILE;
        if (base >= seg_b || base < 0) continue;

        const uint32_t s0_lo = s_lo, s0_hi = s_hi, prev_off0 = prev_off;
        if (base + TILE <= seg_b) {
            /* the common round: 64 steps, then one look at the event */
            const uint8_t *src = tile + tid * ROWB;
            uint4          piece = make_uint4(0, 0, 0, 0);
            uint32_t       av_lo[2][GRP], av_hi[2][GRP], av_kk[2][GRP];
            int32_t        clean_at = -1, clean_how = 0;
            bool           c14 = false;
            auto load_group = [&](int q) {
#pragma unroll
                for (int i = 0; i < GRP; i++) {
                    const int j = q * GRP + i;
                    if ((j & 15) == 0) piece = *reinterpret_cast<const uint4 *>(src + j);
                    const uint32_t word = ((j >> 2) & 3) == 0 ? piece.x : ((j >> 2) & 3) == 1 ? piece.y
                                        : ((j >> 2) & 3) == 2 ? piece.z : piece.w;
                    const uint32_t a = (j & 3) == 0 ? byte_shl<0>(word, sha) : (j & 3) == 1 ? byte_shl<1>(word, sha)
                                     : (j & 3) == 2 ? byte_shl<2>(word, sha) : byte_shl<3>(word, sha);
                    accept_at(a, av_lo[q & 1][i], av_hi[q & 1][i], av_kk[q & 1][i]);
                }
            };
            load_group(0);
#pragma unroll
            for (int q = 0; q < TILE / GRP; q++) {
                if (q + 1 < TILE / GRP) load_group(q + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GRP; i++) {
                    const int      j = q * GRP + i;
                    const uint32_t b_lo = s_lo, b_hi = s_hi;
                    const bool     cl = step(av_lo[q & 1][i], av_hi[q & 1][i], av_kk[q & 1][i]);
                    if ((j & 15) == 14) c14 = cl;
                    /* clean positions are sampled at the end of every 16-byte group */
                    if ((j & 15) == 15 && cl) {
                        const int how = clean_kind(is_snap(b_lo, b_hi), c14, LA && (av_kk[q & 1][i] & 0x8000u) != 0);
                        if (how >= 0) {
                            clean_at = j + 1;
                            clean_how = how;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!event()) {
                if (warm_round) {
                    if (r + 1 == WARM / TILE) s_in = state64();
                } else if (clean_at >= 0) {
                    last_clean = base + clean_at;
                    clean_mode = clean_how;
                }
                continue;
            }
            if (warm_round) {
                /* an event in front of the segment is somebody else's */
                drop_event();
                if (r + 1 == WARM / TILE) s_in = state64();
                continue;
            }
            s_lo = s0_lo;
            s_hi = s0_hi;
            prev_off = prev_off0;
            evv = 0;
        }
        /* byte by byte: a round with an event, or the ragged end of the stream */
        {
            const int64_t end = base + TILE <= seg_b ? base + TILE : seg_b;
            bool          prev_clean = (base == 0);     /* the list at offset 0 is the initial one */
#pragma unroll 1
            for (int64_t p = base; p < end; p++) {
                /* from memory, not from the tile: a 16-byte piece that crosses the end
                 * of the stream is not staged (sre_hip_tile.h) */
                const uint32_t b = data[p];
                const uint32_t b_lo = s_lo, b_hi = s_hi;
                uint32_t       a_lo, a_hi, kk;
                accept_at(b << sha, a_lo, a_hi, kk);
                const bool cl = step(a_lo, a_hi, kk);
                if (event()) {
                    if (!warm_round) {
                        first_ev = p;
                        finished = true;
                        break;
                    }
                    drop_event();
                    prev_clean = false;
                } else if (!warm_round && cl) {
                    const int how = clean_kind(is_snap(b_lo, b_hi), prev_clean, LA && (kk & 0x8000u) != 0);
                    if (how >= 0) {
                        last_clean = p + 1;
                        clean_mode = how;
                    }
                    prev_clean = true;
                } else {
                    prev_clean = false;
                }
            }
            if (warm_round && r + 1 == WARM / TILE) s_in = state64();
        }
    }

    if (!active) return;
    if (LA && last_seg && !finished && !no_eof) {
        /* the extra iteration at end of input (sre_vm_pike.c:235): assertions that hold in front of
         * the end list their continuations; a MATCH among them is an event */
        expand(3u * XCOL);
        if (event()) first_ev = n;
    }
    sre_nfa_summary_t out;
    out.s_in = s_in;
    out.s_out = state64();
    out.first_ev = first_ev;
    out.last_clean = last_clean < 0 ? -1 : last_clean * 2 + clean_mode;
    sum[g] = out;
}

/* ===================================================================== verify */

struct NfaAcc {
    unsigned long long bad, end;        /* init ~0 */
    unsigned long long clean;           /* init 0: 1 + latest clean position in the verified prefix */
};

__global__ __launch_bounds__(256) void
sre_k_nfa_verify_a(sre_scan_geom_t G, const sre_nfa_summary_t *__restrict__ sum, NfaAcc *__restrict__ acc,
                   uint64_t *__restrict__ belief, uint8_t *__restrict__ bvalid)
{
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G.nsegs) return;
    const uint32_t s = nfa_stream_of(G, g);
    const uint64_t k = g - geom_first(G, s);
    if (k > 0) {
        const sre_nfa_summary_t &p = sum[g - 1];
        /* behind a segment that ended the scan nothing is needed */
        if (p.first_ev < 0 && sum[g].s_in != p.s_out) atomicMin(&acc[s].bad, (unsigned long long) k);
        belief[g] = p.s_out;
        bvalid[g] = p.first_ev < 0 ? 1 : 0;
    } else {
        bvalid[g] = 0;
    }
    if (sum[g].first_ev >= 0) atomicMin(&acc[s].end, (unsigned long long) k);
}

__global__ __launch_bounds__(256) void
sre_k_nfa_verify_b(sre_scan_geom_t G, const sre_nfa_summary_t *__restrict__ sum, NfaAcc *__restrict__ acc)
{
    /* nearly every segment has a clean position, and they all go to one address per
     * stream: reduce in the wave, then in the workgroup, then one atomic per 1024 segments
     * (256 threads x 4: a workgroup small enough for a spare slot beside the next scan) */
    __shared__ unsigned long long sh_max;
    const uint64_t g0 = (uint64_t) blockIdx.x * 1024u;
    const uint64_t glast = (g0 + 1023 < G.nsegs) ? g0 + 1023 : G.nsegs - 1;
    const uint32_t s_first = nfa_stream_of(G, g0), s_last = nfa_stream_of(G, glast);
    const bool     uniform = (s_first == s_last);
    if (threadIdx.x == 0) sh_max = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t it = 0; it < 4; it++) {
        const uint64_t g = g0 + it * 256u + threadIdx.x;
        if (g >= G.nsegs) break;
        const uint32_t s = uniform ? s_first : nfa_stream_of(G, g);
        const uint64_t k = g - geom_first(G, s);
        const uint64_t nseg = geom_first(G, s + 1) - geom_first(G, s);
        uint64_t       bad = acc[s].bad, end = acc[s].end;
        if (bad > nseg) bad = nseg;
        if (end > nseg) end = nseg;
        const uint64_t limit = end < bad ? end + 1 : bad;
        if (k < limit && sum[g].last_clean >= 0) {
            const unsigned long long v = (unsigned long long) sum[g].last_clean + 1;    /* position * 2 + mode */
            if (uniform) mine = v > mine ? v : mine;
            else atomicMax(&acc[s].clean, v);
        }
    }
    if (uniform) {
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_down(mine, d, 64);
            mine = o > mine ? o : mine;
        }
        if ((threadIdx.x & 63u) == 0 && mine) atomicMax(&sh_max, mine);
        __syncthreads();
        if (threadIdx.x == 0 && sh_max) atomicMax(&acc[s_first].clean, sh_max);
    }
}

/* per stream: the status word, and the record of every stream that needs no VM
 * window (no event; Thompson) */
__global__ void
sre_k_nfa_verify_c(int mode, sre_scan_geom_t G, const sre_nfa_summary_t *__restrict__ sum,
                   NfaAcc *__restrict__ accs, sre_nfa_status_t *__restrict__ status,
                   int64_t *__restrict__ records, uint32_t ovec_slots, const int64_t *__restrict__ lo)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.nstreams) return;
    /* take this stream's accumulator and leave it reset for the next pass */
    const NfaAcc acc = accs[s];
    accs[s].bad = accs[s].end = ~0ull;
    accs[s].clean = 0;
    if (lo != nullptr && lo[s] < 0) return;     /* settled in an earlier round */
    const uint64_t first = geom_first(G, s), nseg = geom_first(G, s + 1) - first;
    uint64_t       bad = acc.bad, end = acc.end;
    if (bad > nseg) bad = nseg;
    if (end > nseg) end = nseg;
    sre_nfa_status_t st;
    st.first_bad = (int64_t) bad;
    st.done = ((end < bad) || (bad >= nseg)) ? 1 : 0;
    st.ev_pos = (st.done && end < nseg) ? sum[first + end].first_ev : -1;
    st.clean_pos = acc.clean ? (int64_t) ((acc.clean - 1) >> 1) : 0;
    st.clean_mode = acc.clean ? (int32_t) ((acc.clean - 1) & 1) : 0;
    /* ^ in the closure seeded at clean_pos goes by the byte in front of it (find-all rounds restart there) */
    if (st.clean_pos > 0 && st.clean_pos <= (int64_t) geom_len(G, s) && geom_ptr(G, s)[st.clean_pos - 1] == '\n') st.clean_mode |= 2;
    status[s] = st;

    int64_t *rec = records + (size_t) s * (2 + ovec_slots);
    for (uint32_t q = 0; q < ovec_slots; q++) rec[2 + q] = -1;
    if (!st.done) {
        rec[0] = RC_ERROR;
        rec[1] = 0;
    } else if (st.ev_pos < 0) {
        rec[0] = RC_DECLINED;
        rec[1] = 0;
    } else {
        rec[0] = mode == 0 ? 0 : RC_ERROR;      /* Pike: the window kernel fills it in */
        rec[1] = 1;
    }
}

/* ===================================================================== exact entry sets */

/*
 * A program that never forgets (a thread that stays alive from an x far back: `x[^y]*y...`, `x.*y...` over a
 * stream without newlines) defeats the speculation: a lane cannot know from 128 bytes of warm-up that the
 * thread is alive, a fix-up round carries the knowledge ONE segment further, and every round is a pass over
 * everything behind — 64 MiB took 262 143 rounds, 16 s (tools/nfa_never_forgets.py).  The byte step is a
 * union-homomorphism of the thread set (sre_nfa.h), so a segment's effect on ANY entry set follows from its
 * effect on the singletons:
 *      F_k(B u M) = F_k(B) u U_{i in M} F_k({i})
 * sre_k_nfa_seg_matrix walks every unsettled segment with one WAVE — lane i enters with {i} — and stores the 64
 * exit sets (512 bytes a segment; the step written once, generically, for every table form: this is the
 * fallback, 64 walks per segment).  sre_k_nfa_exact_entries then runs, one wave per stream, the recurrence
 *      T_{k+1} = E_k u U_{i in T_k \ B_k} F_k({i})
 * from the verified prefix on (B_k, E_k: what the last pass's lane believed and ended in; beliefs only ever
 * under-estimate) up to the first segment that reports an event, and leaves T_k as every lane's belief: the next
 * pass is exact in every lane, and the chain check proves it (a mistake here costs rounds, never an answer).
 */
struct NfaFnTables {
    int                 use_sa;
    sre_nfa_tables_t    P;
    sre_nfa_sa_tables_t A;
};

__device__ inline uint64_t
nfa_readlane64(uint64_t v, uint32_t i)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, (int) i);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), (int) i);
    return ((uint64_t) hi << 32) | lo;
}

/* one byte step of the set S, as the scan kernels take it (sre_k_nfa / sre_k_nfa_sa); acc = the byte's accept
 * entry, prevk / ck = the kinds of the byte in front and of this one (look-ahead forms) */
__device__ inline uint64_t
nfa_generic_step(const NfaFnTables &T, uint64_t S, uint64_t acc, uint32_t prevk, uint32_t ck)
{
    if (T.use_sa) {
        const sre_nfa_sa_tables_t &A = T.A;
        if (A.nassert) {
            const uint64_t am = (1ull << A.nassert) - 1;
            if (S & am) S |= A.expand[((size_t) (prevk * 4 + ck) << A.nassert) + (size_t) (S & am)];
        }
        const uint64_t t = S & acc;
        uint64_t       e = A.seed;
        for (uint32_t q = 0; q < A.nlut; q++) {
            const uint32_t h = (A.perm >> (8 * q)) & 7u;            /* v_perm_b32 selector: 0-3 low word, 4-7 high (w64) */
            const uint32_t hb = A.w64 ? h : (h & 3u);
            e |= A.lut[(size_t) q * 256 + (size_t) ((t >> (8 * hb)) & 0xffu)];
        }
        const uint64_t ts = A.masked ? (t & A.shift_src) : t;
        uint64_t       sh;
        if (A.w64 && A.carry) sh = ts << 1;
        else sh = (uint64_t) (uint32_t) ((uint32_t) ts << 1) | ((uint64_t) (uint32_t) ((uint32_t) (ts >> 32) << 1) << 32);
        uint64_t S1 = sh | (t & A.self) | e;
        if (!A.w64) S1 &= 0xffffffffull;
        return S1;
    }
    const sre_nfa_tables_t &P = T.P;
    const uint32_t          fsl = P.nassert ? P.nslices - 1 : P.nslices;
    if (P.nassert) S |= P.expand[(size_t) (prevk * 4 + ck) * 256 + (size_t) ((S >> (8 * (P.nslices - 1))) & 0xffu)];
    const uint64_t t = S & (acc | P.match_bits);
    uint64_t       S1 = t & P.match_bits;                           /* sticky */
    for (uint32_t k = 0; k < fsl; k++) S1 |= P.follow[(size_t) k * 256 + (size_t) ((t >> (8 * k)) & 0xffu)];
    return S1;
}

__global__ __launch_bounds__(64) void
sre_k_nfa_seg_matrix(NfaFnTables T, sre_scan_geom_t G, const int64_t *__restrict__ lo, uint64_t *__restrict__ mat)
{
    const uint64_t g = blockIdx.x;
    if (g >= G.nsegs) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t sidx = nfa_stream_of(G, g);
    const uint64_t k = g - geom_first(G, sidx);
    if (lo[sidx] < 0 || (int64_t) k < lo[sidx]) return;
    const uint8_t *data = geom_ptr(G, sidx);
    const int64_t  n = (int64_t) geom_len(G, sidx);
    const int64_t  seg_a = (int64_t) k * G.seg_bytes;
    int64_t        seg_b = seg_a + G.seg_bytes;
    if (seg_b > n) seg_b = n;
    const uint64_t *accept = T.use_sa ? T.A.accept : T.P.accept;
    const uint8_t  *kind = T.use_sa ? T.A.kind : T.P.kind;
    const bool      la = (T.use_sa ? T.A.nassert : T.P.nassert) != 0;
    const uint64_t  valid = T.use_sa ? T.A.valid : (T.P.nbits >= 64 ? ~0ull : ((1ull << T.P.nbits) - 1));
    uint64_t        S = (1ull << lane) & valid;
    uint32_t        prevk = 3;
    if (la && seg_a > 0) prevk = kind[data[seg_a - 1]] & 3u;
    for (int64_t p = seg_a; p < seg_b; p += 64) {
        const int64_t  idx = p + lane;
        const uint32_t byte = idx < seg_b ? data[idx] : 0u;
        const uint64_t acc = accept[byte];
        const uint32_t kd = la ? (kind[byte] & 3u) : 0u;
        const uint32_t nb = seg_b - p < 64 ? (uint32_t) (seg_b - p) : 64u;
        for (uint32_t j = 0; j < nb; j++) {
            const uint32_t ck = (uint32_t) __builtin_amdgcn_readlane((int) kd, (int) j);
            S = nfa_generic_step(T, S, nfa_readlane64(acc, j), prevk, ck);
            prevk = ck;
        }
    }
    mat[g * 64 + lane] = S & valid;
}

__global__ __launch_bounds__(64) void
sre_k_nfa_exact_entries(sre_scan_geom_t G, const sre_nfa_summary_t *__restrict__ sum, const int64_t *__restrict__ lo,
                        const uint64_t *__restrict__ mat, uint64_t *__restrict__ belief, uint8_t *__restrict__ bvalid)
{
    const uint32_t s = blockIdx.x;
    if (s >= G.nstreams || lo[s] < 0) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t first = geom_first(G, s), nseg = geom_first(G, s + 1) - first;
    const uint64_t kstart = (uint64_t) lo[s];
    if (kstart == 0 || kstart >= nseg) return;
    /* T = the exact entry set of the segment at hand, uniform.  The chain is serial in the segments, so nothing
     * on it may wait for memory: the summaries of 64 segments arrive with one load per lane, the singleton exits
     * 16 segments ahead (one row = one coalesced 512-byte load), and the few rows of the threads that were
     * missing from a lane's belief come out of registers by v_readlane. */
    uint64_t T = nfa_readlane64(belief[first + kstart], 0);     /* the verified prefix's exit set (sre_k_nfa_verify_a) */
    uint64_t stop = nseg;                                       /* first segment whose entry set is not needed */
    for (uint64_t k0 = kstart; k0 < nseg && stop == nseg; k0 += 64) {
        const uint64_t me = k0 + lane;
        uint64_t       sin_l = 0, sout_l = 0, myT = 0;
        bool           ev_l = false;
        if (me < nseg) {
            const sre_nfa_summary_t c = sum[first + me];
            sin_l = c.s_in;
            sout_l = c.s_out;
            ev_l = c.first_ev >= 0;
        }
        const uint64_t evm = __builtin_amdgcn_ballot_w64(ev_l);
        uint32_t       done_at = 64;                            /* lanes below it hold an exact entry set */
        for (uint32_t b = 0; b < 4 && done_at == 64; b++) {
            uint64_t rows[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const uint64_t q = k0 + b * 16 + u;
                rows[u] = q < nseg ? mat[(first + q) * 64 + lane] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const uint32_t j = b * 16 + u;
                if (done_at != 64) continue;
                if (k0 + j >= nseg) {
                    done_at = j;
                    continue;
                }
                if (lane == j) myT = T;
                if (((evm >> j) & 1ull) || k0 + j + 1 == nseg) {
                    /* this segment ends the scan (or the stream): nothing behind it is needed */
                    done_at = j + 1;
                    stop = k0 + j + 1;
                    continue;
                }
                uint64_t missing = T & ~nfa_readlane64(sin_l, j), r = 0;
                if (__builtin_popcountll(missing) <= 6) {
                    while (missing) {
                        r |= nfa_readlane64(rows[u], (uint32_t) __builtin_ctzll(missing));
                        missing &= missing - 1;
                    }
                } else {
                    r = ((missing >> lane) & 1ull) ? rows[u] : 0ull;
                    for (int d = 32; d >= 1; d >>= 1) r |= __shfl_xor(r, d, 64);
                    r = nfa_readlane64(r, 0);
                }
                T = nfa_readlane64(sout_l, j) | r;
            }
        }
        if (lane < done_at && me < nseg) {
            belief[first + me] = myT;
            bvalid[first + me] = 1;
        }
    }
    /* behind the first event nothing is needed: those lanes keep their warm-up */
    for (uint64_t q = stop + lane; q < nseg; q += 64) bvalid[first + q] = 0;
}

extern "C" hipError_t
sre_launch_nfa_exact_entries(int use_sa, sre_nfa_tables_t ptab, sre_nfa_sa_tables_t atab, sre_scan_geom_t geom,
                             const sre_nfa_summary_t *d_sum, const int64_t *d_lo, uint64_t *d_mat, uint64_t *d_belief,
                             uint8_t *d_bvalid, hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    NfaFnTables T;
    T.use_sa = use_sa;
    T.P = ptab;
    T.A = atab;
    hipLaunchKernelGGL(sre_k_nfa_seg_matrix, dim3((uint32_t) geom.nsegs), dim3(64), 0, stream, T, geom, d_lo, d_mat);
    hipLaunchKernelGGL(sre_k_nfa_exact_entries, dim3(geom.nstreams), dim3(64), 0, stream, geom, d_sum, d_lo, d_mat, d_belief,
                       d_bvalid);
    return hipGetLastError();
}

typedef void (*nfa_kernel_t)(sre_nfa_tables_t, sre_scan_geom_t, sre_nfa_summary_t *, const int64_t *,
                             const uint64_t *, const uint8_t *);

template <int MODE, bool LA>
nfa_kernel_t
nfa_kernel_slices(uint32_t nslices)
{
    switch (nslices) {
    case 1: return LA ? nullptr : sre_k_nfa<MODE, 1, false>;    /* (assertions take a byte of their own) */
    case 2: return sre_k_nfa<MODE, 2, LA>;
    case 3: return sre_k_nfa<MODE, 3, LA>;
    case 4: return sre_k_nfa<MODE, 4, LA>;
    case 5: case 6: return sre_k_nfa<MODE, 6, LA>;
    default: return sre_k_nfa<MODE, 8, LA>;
    }
}

/* the slice count a variant is compiled for (the host builder rounds the same way) */
uint32_t
nfa_round_slices(uint32_t nslices)
{
    return nslices <= 4 ? (nslices ? nslices : 1) : nslices <= 6 ? 6 : 8;
}

nfa_kernel_t
nfa_kernel(int mode, uint32_t nslices, bool la)
{
    if (la) return mode == 0 ? nfa_kernel_slices<0, true>(nslices) : nfa_kernel_slices<1, true>(nslices);
    return mode == 0 ? nfa_kernel_slices<0, false>(nslices) : nfa_kernel_slices<1, false>(nslices);
}

}  // namespace

/* dynamic LDS of a workgroup: the staging tile and the row descriptors */
static size_t
nfa_dynamic_lds(void)
{
    return (size_t) SRE_SCAN_BLOCK * (2 * SRE_SCAN_ROUND + 16) + (size_t) SRE_SCAN_BLOCK * 16;
}

extern "C" size_t
sre_nfa_lds_bytes(uint32_t nslices, int la)
{
    const uint32_t ns = nfa_round_slices(nslices);
    const size_t   w = ns <= 4 ? 4 : 8;
    return (size_t) 256 * (la ? 2 * w : w) + (size_t) (la ? ns - 1 : ns) * 256 * w + (la ? 16 * 256 * w : 16)
           + nfa_dynamic_lds();
}

extern "C" const char *
sre_nfa_kernel_name(int mode, uint32_t nslices, int la, char *buf, size_t n)
{
    snprintf(buf, n, "sre_k_nfa<%d, %u, %s>", mode == 0 ? 0 : 1, nfa_round_slices(nslices), la ? "true" : "false");
    return buf;
}

extern "C" int
sre_nfa_blocks_per_cu(int mode, uint32_t nslices, int la)
{
    int        n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, nfa_kernel(mode, nslices, la != 0), SRE_SCAN_BLOCK,
                                                                nfa_dynamic_lds());
    if (e != hipSuccess || n < 1) n = 1;
    if (n > 8) n = 8;
    return n;
}

extern "C" hipError_t
sre_launch_nfa_scan(int mode, sre_nfa_tables_t tab, sre_scan_geom_t geom, sre_nfa_summary_t *d_sum,
                    const int64_t *d_lo, const uint64_t *d_belief, const uint8_t *d_bvalid,
                    hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    const uint32_t grid = (uint32_t) ((geom.nsegs + SRE_SCAN_BLOCK - 1) / SRE_SCAN_BLOCK);
    nfa_kernel_t   kern = nfa_kernel(mode, tab.nslices, tab.nassert != 0);
    if (kern == nullptr) return hipErrorInvalidValue;
    if (sre_nfa_lds_bytes(tab.nslices, tab.nassert != 0) > 64 * 1024) {
        /* static + dynamic LDS beyond the default limit of a workgroup */
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) nfa_dynamic_lds());
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SRE_SCAN_BLOCK), nfa_dynamic_lds(), stream, tab, geom, d_sum, d_lo,
                       d_belief, d_bvalid);
    return hipGetLastError();
}

extern "C" size_t
sre_nfa_verify_acc_bytes(uint32_t nstreams)
{
    return (size_t) nstreams * sizeof(NfaAcc);
}

extern "C" hipError_t
sre_nfa_verify_acc_init(void *d_acc, uint32_t nstreams, hipStream_t stream)
{
    hipError_t e = hipMemset2DAsync(d_acc, sizeof(NfaAcc), 0xff, 2 * sizeof(unsigned long long), nstreams, stream);
    if (e != hipSuccess) return e;
    return hipMemset2DAsync(static_cast<char *>(d_acc) + 2 * sizeof(unsigned long long), sizeof(NfaAcc), 0,
                            sizeof(unsigned long long), nstreams, stream);
}

extern "C" hipError_t
sre_launch_nfa_verify(int mode, sre_scan_geom_t geom, const sre_nfa_summary_t *d_sum, void *d_acc,
                      sre_nfa_status_t *d_status, uint64_t *d_belief, uint8_t *d_bvalid,
                      int64_t *d_records, uint32_t ovec_slots, const int64_t *d_lo, hipStream_t stream)
{
    if (geom.nstreams == 0) return hipSuccess;
    NfaAcc        *acc = static_cast<NfaAcc *>(d_acc);
    const uint32_t gseg = (uint32_t) ((geom.nsegs + 255) / 256);
    hipLaunchKernelGGL(sre_k_nfa_verify_a, dim3(gseg), dim3(256), 0, stream, geom, d_sum, acc, d_belief, d_bvalid);
    hipLaunchKernelGGL(sre_k_nfa_verify_b, dim3((uint32_t) ((geom.nsegs + 1023) / 1024)), dim3(256), 0, stream,
                       geom, d_sum, acc);
    hipLaunchKernelGGL(sre_k_nfa_verify_c, dim3((geom.nstreams + 63) / 64), dim3(64), 0, stream, mode, geom,
                       d_sum, acc, d_status, d_records, ovec_slots, d_lo);
    return hipGetLastError();
}

/* ---- the shift-and kernel's variants */

namespace {

template <bool W64, bool CARRY, bool MASKED, bool EVACC, bool LA>
nfa_kernel_t
nfa_sa_kernel_nlut(uint32_t nlut)
{
    switch (nlut) {
    case 0: return reinterpret_cast<nfa_kernel_t>(sre_k_nfa_sa<W64, CARRY, MASKED, EVACC, 0, LA>);
    case 1: return reinterpret_cast<nfa_kernel_t>(sre_k_nfa_sa<W64, CARRY, MASKED, EVACC, 1, LA>);
    case 2: return reinterpret_cast<nfa_kernel_t>(sre_k_nfa_sa<W64, CARRY, MASKED, EVACC, 2, LA>);
    case 3: return reinterpret_cast<nfa_kernel_t>(sre_k_nfa_sa<W64, CARRY, MASKED, EVACC, 3, LA>);
    default: return nullptr;
    }
}

template <bool W64, bool CARRY>
nfa_kernel_t
nfa_sa_kernel_opts(const sre_nfa_sa_tables_t &t)
{
    if (t.nassert) return (t.masked && !t.evacc) ? nfa_sa_kernel_nlut<W64, CARRY, true, false, true>(t.nlut) : nullptr;
    if (t.masked) return t.evacc ? nfa_sa_kernel_nlut<W64, CARRY, true, true, false>(t.nlut) : nfa_sa_kernel_nlut<W64, CARRY, true, false, false>(t.nlut);
    return t.evacc ? nfa_sa_kernel_nlut<W64, CARRY, false, true, false>(t.nlut) : nfa_sa_kernel_nlut<W64, CARRY, false, false, false>(t.nlut);
}

/* the address of a kernel is all that is needed of it here (occupancy, attributes); the launch
 * goes through hipLaunchKernel with the real argument list */
const void *
nfa_sa_kernel(const sre_nfa_sa_tables_t &t)
{
    nfa_kernel_t k;
    if (!t.w64) k = nfa_sa_kernel_opts<false, false>(t);
    else if (t.carry) k = nfa_sa_kernel_opts<true, true>(t);
    else k = nfa_sa_kernel_opts<true, false>(t);
    return reinterpret_cast<const void *>(k);
}

size_t
nfa_sa_dynamic_lds(const sre_nfa_sa_tables_t &t)
{
    /* tile, row descriptors, and with look-ahead assertions the expansion table [16][1 << nassert] */
    return (size_t) SRE_SCAN_BLOCK * SRE_TILE2_ROWB + (size_t) SRE_SCAN_BLOCK * 16
           + (t.nassert ? ((size_t) 16 << t.nassert) * (t.w64 ? 8 : 4) : 0);
}

}  // namespace

extern "C" const char *
sre_nfa_sa_kernel_name(const sre_nfa_sa_tables_t *t, char *buf, size_t n)
{
    snprintf(buf, n, "sre_k_nfa_sa<%s, %s, %s, %s, %u, %s>", t->w64 ? "true" : "false", t->carry ? "true" : "false",
             t->masked ? "true" : "false", t->evacc ? "true" : "false", t->nlut, t->nassert ? "true" : "false");
    return buf;
}

extern "C" int
sre_nfa_sa_blocks_per_cu(const sre_nfa_sa_tables_t *t)
{
    int         n = 0;
    const void *k = nfa_sa_kernel(*t);
    if (k == nullptr) return 1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, SRE_SCAN_BLOCK, nfa_sa_dynamic_lds(*t));
    if (e != hipSuccess || n < 1) n = 1;
    if (n > 8) n = 8;
    return n;
}

extern "C" hipError_t
sre_launch_nfa_sa_scan(sre_nfa_sa_tables_t tab, sre_scan_geom_t geom, sre_nfa_summary_t *d_sum,
                       const int64_t *d_lo, const uint64_t *d_belief, const uint8_t *d_bvalid, hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    const uint32_t grid = (uint32_t) ((geom.nsegs + SRE_SCAN_BLOCK - 1) / SRE_SCAN_BLOCK);
    const void    *kern = nfa_sa_kernel(tab);
    if (kern == nullptr) return hipErrorInvalidValue;
    if (nfa_sa_dynamic_lds(tab) > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) nfa_sa_dynamic_lds(tab));
        if (e != hipSuccess) return e;
    }
    void *args[] = {&tab, &geom, &d_sum, &d_lo, &d_belief, &d_bvalid};
    return hipLaunchKernel(kern, dim3(grid), dim3(SRE_SCAN_BLOCK), args, nfa_sa_dynamic_lds(tab), stream);
}
