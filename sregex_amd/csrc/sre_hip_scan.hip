/*
 * sre_hip_scan.hip — the table-driven segment-parallel scanner (gfx950) and the
 * two utility kernels of the measurement harness.
 *
 * Data layout.  A stream is cut into fixed segments; ONE LANE walks ONE
 * SEGMENT, 64 bytes per round, so a wavefront advances 64 independent segments
 * per step and every input byte is examined by exactly one lane (plus a
 * 128-byte speculative warm-up per segment).  A workgroup of 256 lanes stages
 * its 256 segments through LDS in whole 128-byte lines: one line per row for
 * half a wave per stage, 4 x 16-byte global loads per lane with eight adjacent
 * lanes per line, one stage ahead of its use (tile_fetch).  The fetching lane
 * classifies the bytes and stores ready-made fast-table indices (tile_store);
 * rows are padded so that the per-lane ds_read_b128 of "my row" is
 * bank-conflict free.  The automaton's fast table (one 32-bit word per state x
 * index, below 64 KiB) lives in LDS next to the tile; the per-step work of a
 * lane is one add and one dependent LDS lookup per 8 / BITS input bytes.
 *
 * Exactness.  A lane does not know the true automaton state at the start of
 * its segment; it assumes the state reached after a short warm-up and records
 * what it assumed (s_in) and where it ended (s_out).  sre_k_verify checks the
 * chain s_out[k-1] == s_in[k]; segments behind a broken link are re-run from
 * the exact carried state.  Nothing is reported from an unverified segment.
 *
 * Semantics reproduced here, per reference exec (sre_vm_pike.c:148-689): the
 * search runs until the thread list dies (state DEAD) or end of input (one
 * extra step with the EOF symbol, :235); the LAST match event seen wins
 * (:535-553); in COUNT mode the next search starts at the match end, one byte
 * further after an empty match (:179-196, :624-628).
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "sre_hip_scan.h"
#include "sre_hip_tile.h"

#define SRE_HIP_PIKE_COUNT 2
#define EV_DONE 1
#define EV_POP  2
#define RC_DECLINED (-5)
#define RC_ERROR    (-1)

/* ---- benchmark stream generator: bench/gen-data.pl:9 restated on device ---- */

__global__ void
sre_k_gen_data(uint8_t *__restrict__ dst, uint64_t n, uint64_t body, const uint8_t *__restrict__ tail)
{
    /* 16 bytes per lane; "abccc" has period 5, so lane-local phase = (16*i) % 5 */
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x * 16;
    for (uint64_t base = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 16; base < n;
         base += stride)
    {
        uint8_t  v[16];
        unsigned ph = (unsigned) (base % 5);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint64_t i = base + k;
            uint8_t  c = ph == 0 ? 'a' : ph == 1 ? 'b' : 'c';
            if (i >= body && i < n) c = tail[i - body];
            v[k] = c;
            ph = ph == 4 ? 0 : ph + 1;
        }
        if (base + 16 <= n) {
            *reinterpret_cast<uint4 *>(dst + base) = *reinterpret_cast<uint4 *>(v);
        } else {
            for (int k = 0; k < 16 && base + k < n; k++) dst[base + k] = v[k];
        }
    }
}

extern "C" hipError_t
sre_launch_gen_data(void *d_dst, uint64_t n, uint64_t tail_len, const void *d_tail,
                    hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint64_t lanes = (n + 15) / 16;
    uint32_t block = 256;
    uint64_t grid = (lanes + block - 1) / block;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(sre_k_gen_data, dim3((uint32_t) grid), dim3(block), 0, stream,
                       static_cast<uint8_t *>(d_dst), n, n - tail_len,
                       static_cast<const uint8_t *>(d_tail));
    return hipGetLastError();
}

/* ---- plain streaming read: the box's measured HBM read ceiling ---- */

__global__ void
sre_k_read_ceiling(const uint4 *__restrict__ src, uint64_t nvec, uint32_t *__restrict__ sink)
{
    uint32_t       acc = 0;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    uint64_t       i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    /* 4 independent 16-B loads in flight per lane */
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w) + (c.x ^ c.y ^ c.z ^ c.w)
               + (d.x ^ d.y ^ d.z ^ d.w);
    }
    for (; i < nvec; i += stride) {
        uint4 a = src[i];
        acc += a.x ^ a.y ^ a.z ^ a.w;
    }
    /* keep the loads alive: one (practically never taken) store per lane */
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;
}

extern "C" hipError_t
sre_launch_read_ceiling(const void *d_src, uint64_t n, uint32_t *d_sink, hipStream_t stream)
{
    uint64_t nvec = n / 16;
    if (nvec == 0) return hipSuccess;
    hipLaunchKernelGGL(sre_k_read_ceiling, dim3(SRE_CEILING_GRID), dim3(256), 0, stream,
                       static_cast<const uint4 *>(d_src), nvec, d_sink);
    return hipGetLastError();
}

/* ---- the scanner's staging pattern alone: one row of seg_bytes per lane,
 * TILE bytes of every row per round, the wave's 64 rows fetched as 16-byte
 * pieces one round ahead — no automaton work.  Its rate is the ceiling the
 * access pattern itself allows (rows far apart: part-line requests). ---- */

template <int TILE>
__global__ __launch_bounds__(256) void
sre_k_read_pattern(const uint8_t *__restrict__ src, uint64_t n, uint32_t seg_bytes,
                   uint32_t *__restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t occupancy_pad[];
    const uint32_t tid = threadIdx.x, wbase = tid & ~63u, lane = tid & 63u;
    const uint64_t nsegs = n / seg_bytes;
    const uint64_t row0 = (uint64_t) blockIdx.x * 256 + wbase;
    const uint32_t nrounds = seg_bytes / TILE;
    uint32_t       acc = 0;
    uint4          regs[TILE / 16];
    auto fetch = [&](uint32_t r) {
#pragma unroll
        for (uint32_t i = 0; i < TILE / 16; i++) {
            const uint32_t piece = i * 64 + lane;
            const uint64_t row = row0 + piece / (TILE / 16);
            const uint32_t col = piece % (TILE / 16);
            regs[i] = make_uint4(0, 0, 0, 0);
            if (row < nsegs) {
                regs[i] = *reinterpret_cast<const uint4 *>(src + row * seg_bytes + (uint64_t) r * TILE + col * 16);
            }
        }
    };
    fetch(0);
    for (uint32_t r = 0; r < nrounds; r++) {
#pragma unroll
        for (uint32_t i = 0; i < TILE / 16; i++) acc += regs[i].x ^ regs[i].y ^ regs[i].z ^ regs[i].w;
        if (r + 1 < nrounds) fetch(r + 1);
        /* a little dependent work between rounds, as the consumer has */
        acc = acc * 0x9e3779b1u + r;
    }
    if (acc == 0x9e3779b9u) sink[blockIdx.x % SRE_CEILING_GRID] = acc + occupancy_pad[0];
}

extern "C" hipError_t
sre_launch_read_pattern(const void *d_src, uint64_t n, uint32_t seg_bytes, uint32_t tile,
                        uint32_t lds_bytes, uint32_t *d_sink, hipStream_t stream)
{
    if (seg_bytes == 0 || seg_bytes % 128 != 0 || n < seg_bytes) return hipErrorInvalidValue;
    const uint64_t nsegs = n / seg_bytes;
    const uint32_t grid = (uint32_t) ((nsegs + 255) / 256);
    const uint8_t *p = static_cast<const uint8_t *>(d_src);
    if (tile == 64) {
        hipLaunchKernelGGL(sre_k_read_pattern<64>, dim3(grid), dim3(256), lds_bytes, stream, p, n, seg_bytes, d_sink);
    } else if (tile == 128) {
        hipLaunchKernelGGL(sre_k_read_pattern<128>, dim3(grid), dim3(256), lds_bytes, stream, p, n, seg_bytes, d_sink);
    } else if (tile == 256) {
        hipLaunchKernelGGL(sre_k_read_pattern<256>, dim3(grid), dim3(256), lds_bytes, stream, p, n, seg_bytes, d_sink);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

/* ======================================================================= scan */

namespace {



__device__ inline uint32_t
stream_of(const sre_scan_geom_t &G, uint64_t g)
{
    uint32_t a = 0, b = G.nstreams;
    while (b - a > 1) {
        uint32_t m = (a + b) >> 1;
        if (geom_first(G, m) <= g) a = m; else b = m;
    }
    return a;
}

/* Initial list of a search that starts at sp > 0 on a re-armed context: ^ holds
 * iff the byte in front of it is a newline — for a search after a non-empty
 * match that byte is the match's last one (seen_newline, sre_vm_pike.c:586-601),
 * after an empty match it is the byte the caller skipped (:179-196).  (Without
 * ^ in the program the three initial lists coincide.) */
__device__ inline uint32_t
restart_variant_of(const sre_scan_tables_t &T, uint32_t c)
{
    if (c == '\n') return 1u;
    /* \b / \B at the search start: the context's seen_word (:472-473, 594), ASCII [0-9A-Za-z_] */
    if (T.word_restart && ((c - '0') < 10u || ((c | 32u) - 'a') < 26u || c == '_')) return 3u;
    return 2u;
}

__device__ inline uint32_t
restart_variant(const sre_scan_tables_t &T, const uint8_t *data, int64_t sp)
{
    return restart_variant_of(T, data[sp - 1]);
}

enum : uint32_t {
    F_HAS_EV = 1u, F_LM_VALID = 2u, F_FINISHED = 4u, F_ERROR = 8u, F_UNRESOLVED = 16u, F_SKIP_NEXT = 32u,
    F_IN_PENDING = 64u, F_SP_DIRTY = 128u,
    F_SHADOW = 256u,    /* FIRST: the lane is inside a stable stretch (in a shadow row of the fast table) */
    F_PEND_LAZY = 512u, /* COUNT: the last fast span ended in a FRESH state: a match is pending whose event was not
                           recorded — it ends with that span's last byte (settle() replays the span) */
    F_LZ_GROUP = 1024u  /* ... and that span was a 16-byte group, not a 64-byte round */
};

/* the search a lane is currently following */
struct Walk {
    const sre_scan_tables_t *T;
    const uint8_t           *data;      /* stream base */
    const uint16_t          *tr2;       /* [nstates][ncls + 1] next state | event kind << 8 (LDS) */
    int64_t                  n;         /* stream length */
    uint32_t st;                        /* automaton state */
    /* per-lane booleans as bits of ONE register: kept as separate `bool`s each would
     * live in a scalar register pair (a lane mask), and a dozen of them made the scan
     * loop spill scalar registers (57 v_readlane / v_writelane per 64-byte round) */
    uint32_t fl;
    __device__ inline bool f(uint32_t b) const { return (fl & b) != 0; }
    __device__ inline void set(uint32_t b, bool v) { fl = v ? (fl | b) : (fl & ~b); }
    uint8_t  ev_kind;
    uint32_t ev_state, ev_sym;
    int64_t  ev_pos, ev_sp;
    int64_t  ev_apos;                   /* anchor of that event: known state shortly before it */
    uint32_t ev_astate;
    int64_t  cur_sp;                    /* start of the search in flight, -1 unknown */
    int64_t  warm_lo;                   /* first byte of the lane's warm-up (COUNT: restarts in front of it are not followed) */
    int64_t  anchor_pos;                /* start of the tile round being processed and the */
    uint32_t anchor_state;              /* state there (set by the kernel), -1 none */
    /* F_LM_VALID: FIRST: last event seen; COUNT: last completed match */
    uint32_t lm_state, lm_sym;
    int64_t  lm_pos, lm_sp, lm_apos;
    uint32_t lm_astate;
    int64_t  count;
    int64_t  term_pos;
    /* F_FINISHED, F_ERROR, F_UNRESOLVED; F_SKIP_NEXT: COUNT: the next position to be
     * processed is the byte the caller skips after an empty match; it lies beyond the
     * span that was being processed when the match completed */

    __device__ void complete_match()
    {
        count++;
        fl |= F_LM_VALID;
        lm_state = ev_state;
        lm_sym = ev_sym;
        lm_pos = ev_pos;
        lm_sp = ev_sp;
        lm_apos = ev_apos;
        lm_astate = ev_astate;
        fl &= ~F_HAS_EV;
    }
};

/*
 * Exact, byte-at-a-time: process positions [p, p_to) — position n is the EOF
 * step — following every restart the reference's caller would make.  `p` may
 * move backwards (a search restarts at its predecessor's match end).
 * warm: speculative warm-up; nothing is counted and a dead list is replaced by
 * state `warm_seed`.
 */
template <int MODE>
__device__ void slow_run(Walk &w, int64_t p, int64_t p_to, bool warm, uint32_t warm_seed)
{
    const sre_scan_tables_t &T = *w.T;
    const uint32_t           nsym = T.ncls + 1;
    /* the exact path read one byte per step from global memory — a dependent L2 round trip, 0.4 us
     * per byte-step, which is what a stream of look-ahead matches ran at (bench.py `floor`).  The
     * bytes now come eight at a time through a window in registers (round 3). */
    int64_t  win_at = -1;       /* aligned offset of the window, -1: empty */
    uint64_t win_bytes = 0;
    auto byte_at = [&](int64_t q) -> uint32_t {
        const int64_t a = q & ~(int64_t) 7;
        if (a != win_at) {
            win_at = a;
            if (a + 8 <= w.n && ((reinterpret_cast<uintptr_t>(w.data) + (uint64_t) a) & 7u) == 0) {
                win_bytes = *reinterpret_cast<const uint64_t *>(w.data + a);
            } else {
                /* the ragged end of the stream, or a stream that does not start on an 8-byte boundary */
                win_bytes = 0;
                for (int64_t k = 0; k < 8 && a + k < w.n; k++) win_bytes |= (uint64_t) w.data[a + k] << (8 * k);
            }
        }
        return (uint32_t) (win_bytes >> (8 * (q & 7))) & 0xffu;
    };

    if (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT)) {
        w.fl &= ~F_SKIP_NEXT;
        p++;                        /* w.st is already the list of the search that starts behind it */
    }
    while (p < p_to && !w.f(F_FINISHED)) {
        if (p == w.n && ((T.state_flags[w.st] >> 1) & 3) == 2) {
            /* the leading-byte skip ran to the end of input: the reference
             * leaves its loop without the EOF step and returns with threads
             * still listed, which poisons the context (sre_vm_pike.c:304-306,
             * 616-622): a pending match is returned, the next exec fails */
            w.term_pos = p;
            if (MODE == SRE_HIP_PIKE_COUNT) {
                if (w.f(F_HAS_EV)) {
                    w.complete_match();
                    w.fl |= F_ERROR;
                }
            } else if (w.f(F_HAS_EV) || (T.state_flags[w.st] & 1)) {
                /* a speculative lane may not have seen the event itself; the
                 * state says that a match is pending */
                w.fl |= F_ERROR;
            }
            w.fl |= F_FINISHED;
            return;
        }
        const uint32_t         sym = p < w.n ? T.cls[byte_at(p)] : T.ncls;
        /* all the exact step needs of a transition: where it goes and what it reports */
        const uint32_t tr2 = w.tr2[w.st * nsym + sym];
        struct { uint32_t next; uint8_t kind; } tr = {tr2 & 0xffu, (uint8_t) (tr2 >> 8)};
        if (tr.kind) {
            w.fl |= F_HAS_EV;
            w.ev_kind = tr.kind;
            w.ev_state = w.st;
            w.ev_sym = sym;
            w.ev_pos = p;
            w.ev_sp = w.cur_sp;
            /* the round's entry state is a usable anchor if it belongs to this
             * same search and lies at most one round in front of the event */
            /* (strictly behind the search start: a search that a folded match started AT the round's first byte
             * — the byte is read again, sre_scan_host.cpp — does not begin in the round's entry state; found by
             * the random campaign, `\ba+` seed 403) */
            if (w.anchor_pos >= 0 && w.anchor_pos <= p && p - w.anchor_pos <= 256
                && (w.cur_sp < 0 || w.anchor_pos > w.cur_sp))
            {
                w.ev_apos = w.anchor_pos;
                w.ev_astate = w.anchor_state;
            } else {
                w.ev_apos = -1;
                w.ev_astate = 0;
            }
            if (MODE != SRE_HIP_PIKE_COUNT && !warm) {
                w.fl |= F_LM_VALID;
                w.lm_state = w.st;
                w.lm_sym = sym;
                w.lm_pos = p;
                w.lm_sp = w.cur_sp;
                w.lm_apos = w.ev_apos;
                w.lm_astate = w.ev_astate;
            }
        }
        /* Thompson (sre_vm_thompson.c:233-235) answers at the first MATCH thread it meets: the
         * search ends with its first event */
        w.st = (tr.kind && T.mode == 0 && !warm) ? 0u : tr.next;
        if (w.st != 0) {
            p++;
            continue;
        }

        /* the thread list died: this search is over */
        if (warm) {
            if (MODE == SRE_HIP_PIKE_COUNT) {
                /* COUNT: the warm-up follows the caller's restarts like the exact path below (nothing is
                 * counted): where the next search starts — at the match end, a byte further after an
                 * empty match, possibly beyond this span (F_SKIP_NEXT: part of what the lane believes about
                 * its entry) — and from which initial list (the byte in front) is what the lane must get
                 * right.  With a fixed seed every restart of `\b` or `^x` resumed from the wrong list, the
                 * chain check failed at nearly every segment and the fix-up rounds (one verified segment
                 * each) were linear in the stream: 64 MiB of "ab cd " took 131 086 rounds = 151 s. */
                if (w.f(F_HAS_EV)) {
                    const bool    pop = (w.ev_kind == EV_POP || w.ev_kind == SRE_DEV_EV_POP_FULL);
                    const bool    empty = w.ev_kind == EV_POP || w.ev_kind == SRE_DEV_EV_DONE_EMPTY;
                    const int64_t e = pop ? w.ev_pos : w.ev_pos + 1;
                    const int64_t sp = empty ? e + 1 : e;
                    if (e >= w.warm_lo && sp >= 1 && !(empty && e >= w.n)) {
                        w.fl &= ~F_HAS_EV;
                        w.cur_sp = sp;
                        w.st = T.init[restart_variant_of(T, byte_at(sp - 1))];
                        p = sp;
                        if (p > p_to && p_to <= w.n) w.fl |= F_SKIP_NEXT;
                        continue;
                    }
                }
                /* the lane's guess was off (a search only ends with a match or the input): the list a
                 * search behind this byte would start from */
                w.st = p < w.n ? T.init[restart_variant_of(T, byte_at(p))] : warm_seed;
            } else {
                w.st = warm_seed;
            }
            w.fl &= ~F_HAS_EV;
            w.cur_sp = -1;
            p++;
            continue;
        }
        if (MODE != SRE_HIP_PIKE_COUNT) {
            w.term_pos = p;
            w.fl |= F_FINISHED;
            return;
        }
        if (!w.f(F_HAS_EV)) {
            /* no match: only possible at end of input => DECLINED ends the
             * iteration.  Earlier it means this lane assumed a matched-mode
             * state whose event it never saw: it cannot resolve the restart. */
            if (p < w.n) w.fl |= F_UNRESOLVED;
            w.term_pos = p;
            w.fl |= F_FINISHED;
            return;
        }
        {
            /* next exec: from the match end; one byte further after an empty
             * match (sre_vm_pike.c:179-196, 624-628) */
            const bool    pop = (w.ev_kind == EV_POP || w.ev_kind == SRE_DEV_EV_POP_FULL);
            const bool    empty = w.ev_kind == EV_POP || w.ev_kind == SRE_DEV_EV_DONE_EMPTY;
            const int64_t e = pop ? w.ev_pos : w.ev_pos + 1;
            w.complete_match();
            if (empty) {
                if (e >= w.n) {
                    w.term_pos = p;
                    w.fl |= F_FINISHED;      /* size == 0 && eof => DECLINED */
                    return;
                }
                w.cur_sp = e + 1;
            } else {
                w.cur_sp = e;
            }
            w.st = T.init[restart_variant_of(T, byte_at(w.cur_sp - 1))];
            p = w.cur_sp;
            /* an empty match that ended with a consumed byte can put the skipped
             * byte just outside this span */
            if (p > p_to && p_to <= w.n) w.fl |= F_SKIP_NEXT;
            w.anchor_pos = -1;      /* the round's entry state belonged to the previous search */
        }
    }
}

/* COUNT: matches completed inside a pure-fast span (a 16-byte group or a whole
 * round, entered in state s0 with the search in flight started at sp0) were
 * only counted; recover the last one and the start of the search in flight
 * after the span.  With at least one match in the span the latter does not
 * depend on sp0.  Deliberately a real call, by value: it runs a few times per
 * segment and must not add to the scan loop's register pressure. */
struct SpanResult {
    int64_t  sp;                /* start of the search in flight after the span */
    int64_t  last_pos, last_sp; /* last completed match: its event's byte (-1 none), its search start */
    uint32_t last_state, last_sym;
    int64_t  pend_pos;          /* the match pending at the end of the span, recorded inside it (-1 none) */
    uint32_t pend_state, pend_sym;
};

/* pend0_*: the match pending in front of the span (its event; pend0_pos < 0: none, or unknown and superseded) */
__device__ __attribute__((noinline)) SpanResult
resolve_fast_span(const sre_scan_tables_t *Tp, const uint16_t *tr2, const uint8_t *data, int64_t gpos,
                  uint32_t len, uint32_t s0, int64_t sp0, int64_t pend0_pos, uint32_t pend0_state, uint32_t pend0_sym)
{
    const sre_scan_tables_t &T = *Tp;
    SpanResult r;
    uint32_t   st = s0;
    r.sp = sp0;
    r.last_pos = r.last_sp = -1;
    r.last_state = r.last_sym = 0;
    r.pend_pos = pend0_pos;
    r.pend_state = pend0_state;
    r.pend_sym = pend0_sym;
    for (uint32_t b = 0; b < len; b++) {
        const uint32_t         sym = T.cls[data[gpos + b]];
        const uint32_t t2 = tr2[st * (T.ncls + 1) + sym];
        struct { uint32_t next; uint8_t kind; } tr = {t2 & 0xffu, (uint8_t) (t2 >> 8)};
        if (tr.kind && tr.next == 0) {
            /* a folded match: behind this byte, or (a look-ahead assertion completed it) in front of it —
             * then the next search reads the byte again, without an event (sre_scan_host.cpp) */
            const bool pop = tr.kind == SRE_DEV_EV_POP_FULL && gpos + b > 0;
            r.last_pos = gpos + b;
            r.last_state = st;
            r.last_sym = sym;
            r.last_sp = r.sp;
            r.sp = pop ? gpos + b : gpos + b + 1;
            r.pend_pos = -1;
            st = T.init[restart_variant(T, data, r.sp)];
            if (pop) st = tr2[st * (T.ncls + 1) + sym] & 0xffu;
        } else if (tr.kind) {
            /* the pending match grows (a FRESH state follows) */
            r.pend_pos = gpos + b;
            r.pend_state = st;
            r.pend_sym = sym;
            st = tr.next;
        } else if (tr.next == 0 && gpos + b > 0) {
            /* the list dies in a FRESH state: the pending match, which ends in front of this byte, completes;
             * the next search reads the byte again — and may record a match with it */
            r.last_pos = r.pend_pos;            /* (unknown only for a completion that a later one supersedes) */
            r.last_state = r.pend_state;
            r.last_sym = r.pend_sym;
            r.last_sp = r.sp;
            r.sp = gpos + b;
            st = T.init[restart_variant(T, data, r.sp)];
            const uint32_t t3 = tr2[st * (T.ncls + 1) + sym];
            r.pend_pos = -1;
            if ((t3 >> 8) && (t3 & 0xffu) == 0) {
                /* ... and completes at once (the byte itself, or an empty match in front of it that makes the
                 * caller skip the byte): the search behind it starts at the next byte */
                r.last_pos = gpos + b;
                r.last_state = st;
                r.last_sym = sym;
                r.last_sp = r.sp;
                r.sp = gpos + b + 1;
                st = T.init[restart_variant(T, data, r.sp)];
            } else {
                if (t3 >> 8) {
                    r.pend_pos = gpos + b;
                    r.pend_state = st;
                    r.pend_sym = sym;
                }
                st = t3 & 0xffu;
            }
        } else {
            st = tr.next;
        }
    }
    return r;
}

/* (a & 0xffff) + ((b >> 16 H) & 0xffff) in ONE instruction (H == 2: + b): the SDWA
 * operand selects feed the adder, so a table lookup costs one address op */
template <int H>
__device__ inline uint32_t
add_w0_w(uint32_t a, uint32_t b)
{
    uint32_t r;
    if (H == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    if (H == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    if (H == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* a + (b >> 24): the COUNT field of a fast-table entry */
__device__ inline uint32_t
add_b3(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

/* ((v >> 8K) & 255) << sh */
template <int K>
__device__ inline uint32_t
byte_x4(uint32_t v, uint32_t sh)
{
    uint32_t r;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh), "v"(v));
    if (K == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh), "v"(v));
    return r;
}

/* the kernels behind a scan (chain check, capture walk) are a few small waves that usually run
 * while the NEXT scan holds the chip: first in line at the issue arbiter */
#ifdef SRE_NO_TAIL_PRIO
#define SRE_TAIL_PRIO() ((void) 0)
#else
#define SRE_TAIL_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

#ifndef SRE_SCAN_PREFETCH
#define SRE_SCAN_PREFETCH 1         /* stages of HBM loads in flight per lane (1 or 2), see sre_k_scan */
#endif
#ifndef SRE_FIRST_BLOCKS
#define SRE_FIRST_BLOCKS 4          /* workgroups per CU the FIRST / Thompson kernels' registers are budgeted for */
#endif
#ifndef SRE_COUNT_BLOCKS
#define SRE_COUNT_BLOCKS 3          /* workgroups per CU the COUNT kernel's registers are budgeted for */
#endif

/*
 * BITS = class bits per input byte: one fast-table lookup advances 8 / BITS
 * bytes (BITS == 8: the index is the byte itself).  A lane consumes
 * SRE_SCAN_ROUND (64) bytes per round.  Rounds 0 and 1 are the speculative
 * warm-up: the 128 bytes (one line) in front of the segment, walked with the
 * same fast loop, nothing recorded.
 */
/* GROW: COUNT tables with FRESH states (growing matches on the fast path, sre_scan_host.cpp): the lane keeps the
 * last fast span to recover the unrecorded pending match.  A variant of its own: the bookkeeping cost the tables
 * without such states 5 % (configs[2], same box) when it was a run-time switch. */
template <int MODE, int BITS, bool WIDE, bool GROW>
__global__ __launch_bounds__(SRE_SCAN_BLOCK, MODE == SRE_HIP_PIKE_COUNT ? SRE_COUNT_BLOCKS : SRE_FIRST_BLOCKS) void
sre_k_scan(const sre_scan_tables_t *__restrict__ tabp, sre_scan_geom_t G,
           sre_seg_summary_t *__restrict__ sum, const sre_stream_status_t *__restrict__ st_lo,
           const uint8_t *__restrict__ entry)
{
    constexpr int      TILE = SRE_SCAN_ROUND;
    constexpr int      WARM = SRE_SCAN_LINE;         /* warm-up bytes in front of a segment */
    constexpr int      STRIDE = 8 / BITS;
    constexpr uint32_t ROWRAW = TILE / STRIDE * (WIDE ? 2 : 1);  /* index bytes per round */
    constexpr uint32_t ROWB = 2 * ROWRAW + 16;       /* see tile_store */
    constexpr int      GIDX = 16 / STRIDE;           /* indices per 16 input bytes */
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    /* class map pre-shifted for each byte position of an index, scaled by 4 */
    __shared__ uint16_t clsx[BITS == 8 ? 1 : 8 / BITS][256];
    __shared__ sre_scan_tables_t Ts;

    const uint32_t tid = threadIdx.x;
    /* LDS: [fast table: states, trap row, shadow rows][class map 256][transitions, 2 B each]
     *      [state flags][tile] */
    const uint32_t nst = tabp->nstates, nrows = tabp->fast_rows, nsh = tabp->nshadow;
    uint32_t *fast = reinterpret_cast<uint32_t *>(lds);
    uint8_t  *clsl = lds + nrows * SRE_FAST_ROW_BYTES;
    uint8_t  *trl = clsl + 256;
    const uint32_t tr_bytes = (nst * (tabp->ncls + 1) * 2u + 15u) & ~15u;      /* compact: 2 bytes a transition */
    uint8_t  *sfl = trl + tr_bytes;
    uint8_t  *tile = lds + (((size_t) (sfl - lds) + nst + 15u) & ~(size_t) 15u);
    /* (the row descriptors live behind the tile, not in static LDS: the fast table is
     * addressed with 16 bits and every static kilobyte in front of it costs a state) */
    RowDesc  *rows = reinterpret_cast<RowDesc *>(tile + SRE_SCAN_BLOCK * ROWB);
    (void) nrows;
    if (tid == 0) {
        Ts = *tabp;
        /* the exact path reads its tables from LDS too */
        Ts.cls = clsl;
        Ts.state_flags = sfl;
    }
    /* In LDS an entry is [matches completed : 8][flags : 8][LDS byte address of the
     * next state's row : 16] (the fast table lies below 64 KiB): a lookup address
     * is then one add of two 16-bit fields and the match count one add of a byte
     * field, both selected by the add instruction itself.  SLOW entries point to
     * the TRAP row, whose entries all point back to it with the flag set: a chain
     * of lookups carries no per-step flag test, the flag is there at its end. */
    constexpr uint32_t LDS_SLOW = 1u << 16;
    constexpr uint32_t LDS_CNT_SHIFT = 24;
    constexpr uint32_t LDS_FRESH = 1u << 17;        /* COUNT: the entry ends in a FRESH state */
    const uint32_t fast_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) lds;
    const uint32_t trap_lds = fast_lds + nst * SRE_FAST_ROW_BYTES;
    const uint32_t shadow_lds = trap_lds + SRE_FAST_ROW_BYTES;      /* first shadow row */
    auto to_lds = [fast_lds, trap_lds](uint32_t g) -> uint32_t {
        if (g & SRE_FAST_SLOW) return trap_lds | LDS_SLOW;
        /* COUNT: the count byte also carries the entry's EVT flag (bit 7): one SDWA add per lookup sums
         * both — completions in the low seven bits (<= 64 a round), entries with a growing match above */
        return (fast_lds + (g & ~(SRE_FAST_ROW_BYTES - 1)))
               | (((g >> SRE_FAST_CNT_SHIFT) & SRE_FAST_CNT_MASK) << LDS_CNT_SHIFT)
               | ((MODE == SRE_HIP_PIKE_COUNT && (g & SRE_FAST_EVT)) ? (128u << LDS_CNT_SHIFT) : 0u)
               | ((MODE == SRE_HIP_PIKE_COUNT && (g & SRE_FAST_NEXT_FRESH)) ? LDS_FRESH : 0u);
    };
    for (uint32_t i = tid; i < nst * 64; i += SRE_SCAN_BLOCK) {
        uint4 e = reinterpret_cast<const uint4 *>(tabp->fast)[i];
        e.x = to_lds(e.x);
        e.y = to_lds(e.y);
        e.z = to_lds(e.z);
        e.w = to_lds(e.w);
        reinterpret_cast<uint4 *>(fast)[i] = e;
    }
    fast[nst * 256 + tid] = trap_lds | LDS_SLOW;
    for (uint32_t q = 0; q < nsh; q++) {
        /* shadow of state s: its STABLE entries stay in the copy, all others leave it */
        const uint32_t s = tabp->shadow_state[q], addr = shadow_lds + q * SRE_FAST_ROW_BYTES;
        const uint32_t gl = tabp->fast[s * 256 + tid];
        fast[(nst + 1 + q) * 256 + tid] = (gl & SRE_FAST_STABLE) ? addr : to_lds(gl);
    }
    for (uint32_t i = tid; i < nst * (tabp->ncls + 1); i += SRE_SCAN_BLOCK) {
        reinterpret_cast<uint16_t *>(trl)[i] = tabp->trans2[i];
    }
    for (uint32_t i = tid; i < tabp->nstates; i += SRE_SCAN_BLOCK) sfl[i] = tabp->state_flags[i];
    clsl[tid] = tabp->cls[tid];
    if (BITS != 8) {
#pragma unroll
        for (int u = 0; u < 8 / BITS; u++) clsx[u][tid] = (uint16_t) ((uint32_t) tabp->cls[tid] << (BITS * u + (WIDE ? 2 : 0)));
    }
    __syncthreads();
    const sre_scan_tables_t &T = Ts;

    /* ---- which segment am I ---- */
    const uint64_t g = (uint64_t) blockIdx.x * SRE_SCAN_BLOCK + tid;
    bool           active = g < G.nsegs;
    uint32_t       sidx = 0;
    uint64_t       k = 0;                   /* segment index inside the stream */
    int64_t        lo_s = -2;               /* first segment of this stream that is re-run (-2: first pass) */
    if (active) {
        uint32_t a = 0, b = G.nstreams;     /* seg_first[a] <= g < seg_first[b] */
        while (b - a > 1) {
            uint32_t m = (a + b) >> 1;
            if (geom_first(G, m) <= g) a = m; else b = m;
        }
        sidx = a;
        k = g - geom_first(G, a);
        /* a fix-up round: the streams that are not settled yet, from their first wrong
         * segment on — read from the status words of the previous round, on the device */
        if (st_lo != nullptr) {
            lo_s = st_lo[sidx].done ? -1 : st_lo[sidx].first_bad;
            if (lo_s < 0 || (int64_t) k < lo_s) active = false;
        }
    }

    Walk w;
    w.T = &T;
    w.tr2 = reinterpret_cast<const uint16_t *>(trl);
    w.data = nullptr;
    w.n = 0;
    w.st = 0;
    w.ev_kind = 0;
    w.ev_state = w.ev_sym = 0;
    w.ev_pos = w.ev_sp = -1;
    w.ev_apos = -1;
    w.ev_astate = 0;
    w.cur_sp = -1;
    w.warm_lo = 0;
    w.anchor_pos = -1;
    w.anchor_state = 0;
    w.lm_state = w.lm_sym = 0;
    w.lm_pos = w.lm_sp = w.lm_apos = -1;
    w.lm_astate = 0;
    w.count = 0;
    w.term_pos = -1;
    w.fl = 0;

    int64_t  seg_a = 0, seg_b = 0;
    uint32_t s_in = 0, seed = 0;
    /* COUNT: the pending match this lane enters its segment with (see sre_seg_summary_t) */
    int64_t  in_pe_pos = -1;
    uint32_t in_pe_state = 0, in_pe_sym = 0;
    bool     last_seg = false, warm = false;
    RowDesc mine;
    mine.addr = 0;
    mine.lo = 0;
    mine.hi16 = -1;                         /* nothing readable */
    if (active) {
        w.data = geom_ptr(G, sidx);
        w.n = (int64_t) geom_len(G, sidx);
        seg_a = (int64_t) k * G.seg_bytes;
        seg_b = seg_a + G.seg_bytes;
        const uint64_t nseg = geom_first(G, sidx + 1) - geom_first(G, sidx);
        last_seg = (k + 1 == nseg);
        if (seg_b > w.n) seg_b = w.n;

        if (k == 0) {
            /* a chunk of a stream whose search is already under way enters with the
             * state the previous chunk ended in (sre_k_stream_tail) */
            w.st = (G.flags & SRE_GEOM_CONTINUES) ? G.entry_state : T.init[G.init_variant];
            w.cur_sp = 0;
        } else if (lo_s >= 0 && (int64_t) k == lo_s) {
            /* exact carry from the verified predecessor */
            const sre_seg_summary_t &c = sum[g - 1];
            w.st = c.s_out & ~SRE_STATE_SKIP;
            w.set(F_SKIP_NEXT, MODE == SRE_HIP_PIKE_COUNT && (c.s_out & SRE_STATE_SKIP) != 0);
            w.cur_sp = c.cur_sp;
            if (c.flags & SRE_SUM_PENDING) {
                w.fl |= F_HAS_EV;
                w.ev_state = c.pe_state;
                w.ev_sym = c.pe_sym;
                w.ev_pos = c.pe_pos;
                w.ev_sp = c.pe_sp;
                w.ev_apos = c.pe_apos;
                w.ev_astate = c.pe_astate;
                w.ev_kind = (uint8_t) (w.tr2[c.pe_state * (T.ncls + 1) + c.pe_sym] >> 8);
                if (MODE == SRE_HIP_PIKE_COUNT) {
                    w.fl |= F_IN_PENDING;
                    in_pe_pos = c.pe_pos;
                    in_pe_state = c.pe_state;
                    in_pe_sym = c.pe_sym;
                }
            }
        } else if (entry != nullptr && entry[g] != 0xffu) {
            /* exact: the segments' transition functions were composed up to here
             * (sre_k_seg_functions; FIRST / Thompson after fix-up rounds that did
             * not converge) */
            w.st = entry[g];
            if (MODE == SRE_HIP_PIKE_COUNT && lo_s > 0 && (T.state_flags[w.st] & 1)) {
                /* (COUNT: the match such a state holds is taken from the verified prefix, as below) */
                const sre_seg_summary_t &c = sum[geom_first(G, sidx) + lo_s - 1];
                if (c.flags & SRE_SUM_PENDING) {
                    w.fl |= F_HAS_EV | F_IN_PENDING;
                    w.ev_state = c.pe_state;
                    w.ev_sym = c.pe_sym;
                    w.ev_pos = c.pe_pos;
                    w.ev_sp = c.pe_sp;
                    w.ev_apos = -1;
                    w.ev_astate = 0;
                    w.ev_kind = (uint8_t) (w.tr2[c.pe_state * (T.ncls + 1) + c.pe_sym] >> 8);
                    in_pe_pos = c.pe_pos;
                    in_pe_state = c.pe_state;
                    in_pe_sym = c.pe_sym;
                }
            }
        } else {
            /* speculative: assume the state reached by a warm-up over the WARM
             * bytes in front of the segment.  In a fix-up round the warm-up
             * starts from the state the verified prefix ended in (it tends to
             * recur), otherwise from the initial state. */
            warm = true;
            seed = T.init[0];
            if (lo_s > 0) {
                const sre_seg_summary_t &c = sum[geom_first(G, sidx) + lo_s - 1];
                const uint32_t           cs = c.s_out & ~SRE_STATE_SKIP;
                if (cs != 0) {
                    if (!(MODE == SRE_HIP_PIKE_COUNT && (T.state_flags[cs] & 1))) {
                        seed = cs;
                    } else if (c.flags & SRE_SUM_PENDING) {
                        /* COUNT cannot resolve a pending match it has not seen — but it
                         * can take over the one the verified prefix ends with: a match
                         * whose list lives on for long (a.*b with no b in sight) is then
                         * believed by every lane behind it, and the chain check compares
                         * exactly that belief */
                        seed = cs;
                        w.fl |= F_HAS_EV;
                        w.ev_state = c.pe_state;
                        w.ev_sym = c.pe_sym;
                        w.ev_pos = c.pe_pos;
                        w.ev_sp = c.pe_sp;
                        w.ev_apos = -1;
                        w.ev_astate = 0;
                        w.ev_kind = (uint8_t) (w.tr2[c.pe_state * (T.ncls + 1) + c.pe_sym] >> 8);
                    }
                }
            }
            w.st = seed;
        }
        w.warm_lo = seg_a >= (int64_t) WARM ? seg_a - (int64_t) WARM : 0;
        s_in = w.st | (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT) ? SRE_STATE_SKIP : 0u);   /* what the chain check compares */
        /* row = [seg_a - WARM, seg_b): the warm-up rounds, then the segment */
        mine.addr = (uint64_t) reinterpret_cast<uintptr_t>(w.data) + (uint64_t) (seg_a - WARM);
        /* the warm-up of a stream's second segment may be cut short by the stream start */
        mine.lo = warm ? (seg_a >= WARM ? 0 : (int32_t) (WARM - seg_a)) : WARM;
        mine.hi16 = (int32_t) (WARM + (seg_b - seg_a)) - 16;
    }
    rows[tid] = mine;

    /* pure-fast COUNT bookkeeping: the last 16-byte group that completed matches
     * without leaving the fast loop, while it still holds the segment's last
     * completed match */
    int64_t  fcA_pos = -1, fcB_pos = -1, fc_sp0 = -1;
    uint32_t fcA_len = 0, fcA_s0 = 0, fcB_len = 0, fcB_s0 = 0;
    /* ... and the last fast span that ended in a FRESH state (F_PEND_LAZY): where it began and in which state */
    int64_t  lz_pos = -1;
    uint32_t lz_s0 = 0;
    /* a pure-fast span [pos, pos + len), entered in state s0, completed cnt matches */
    auto note_span = [&](int64_t pos, uint32_t len, uint32_t s0, uint32_t cnt, bool warm_round) {
        w.fl &= ~F_HAS_EV;                       /* superseded */
        if (warm_round) return;
        if (w.f(F_SP_DIRTY)) {
            fcA_pos = fcB_pos;
            fcA_len = fcB_len;
            fcA_s0 = fcB_s0;
        } else {
            fc_sp0 = w.cur_sp;                  /* exact right now */
            fcA_pos = -1;
        }
        fcB_pos = pos;
        fcB_len = len;
        fcB_s0 = s0;
        if (w.f(F_PEND_LAZY)) {
            /* the span in front ended with an unrecorded pending match, which this one may complete with its
             * first byte: the replay starts there */
            fcB_pos = lz_pos;
            fcB_len = len + (w.f(F_LZ_GROUP) ? 16u : (uint32_t) TILE);
            fcB_s0 = lz_s0;
        }
        w.fl |= F_SP_DIRTY;
        w.count += cnt;
    };
    constexpr bool any_fresh = GROW;
    /* after a fast span of a COUNT scan: sum = what the lookups' count bytes added up to (completions in the
     * low seven bits, entries with a growing match above), fresh1 = it ended in a FRESH state (its last entry says so) */
    auto fast_span_done = [&](int64_t pos, uint32_t len, uint32_t s0, uint32_t sum, bool fresh1, bool warm_round) {
        if (!any_fresh) {
            if (sum) note_span(pos, len, s0, sum, warm_round);      /* (a table without growing matches: completions only) */
            return;
        }
        if (sum == 0) {
            w.fl &= ~F_PEND_LAZY;               /* (a FRESH state is entered by a recorded match only) */
            return;
        }
        if (sum & 127u) note_span(pos, len, s0, sum & 127u, warm_round);
        else if (sum) w.fl &= ~F_HAS_EV;        /* a later match of the same search supersedes the recorded one */
        if (fresh1) {
            w.fl |= F_PEND_LAZY;
            w.set(F_LZ_GROUP, len == 16u);
            lz_pos = pos;
            lz_s0 = s0;
        } else {
            w.fl &= ~F_PEND_LAZY;
        }
    };
    /* make w.cur_sp exact again, record the last completed match (lm_*) and the pending one (ev_*) */
    auto settle = [&]() {
        if (MODE != SRE_HIP_PIKE_COUNT || !(w.fl & (F_SP_DIRTY | F_PEND_LAZY))) return;
        if (w.f(F_SP_DIRTY)) {
            int64_t sp0 = fc_sp0;
            if (fcA_pos >= 0) sp0 = resolve_fast_span(&T, w.tr2, w.data, fcA_pos, fcA_len, fcA_s0, -1, -1, 0, 0).sp;
            /* (a span that begins in a FRESH state and was not extended begins right behind the exact path: the
             * pending match is the one that path recorded) */
            const SpanResult r = resolve_fast_span(&T, w.tr2, w.data, fcB_pos, fcB_len, fcB_s0, sp0, w.ev_pos, w.ev_state, w.ev_sym);
            if (r.last_pos >= 0) {
                w.fl |= F_LM_VALID;
                w.lm_pos = r.last_pos;
                w.lm_state = r.last_state;
                w.lm_sym = r.last_sym;
                w.lm_sp = r.last_sp;
                /* the span's entry state is the capture walker's anchor when the match's search was under way
                 * there (strictly: see slow_run) — without one it replays the segment up to the event */
                const bool anchored = r.last_sp < fcB_pos && r.last_pos - fcB_pos <= 256 && fcB_pos >= seg_a;    /* (never in the warm-up) */
                w.lm_apos = anchored ? fcB_pos : (int64_t) -1;
                w.lm_astate = anchored ? fcB_s0 : 0u;
            }
            w.cur_sp = r.sp;
            w.fl &= ~F_SP_DIRTY;
        }
        if (w.f(F_PEND_LAZY)) {
            const SpanResult r = resolve_fast_span(&T, w.tr2, w.data, lz_pos, w.f(F_LZ_GROUP) ? 16u : (uint32_t) TILE, lz_s0,
                                                   -1, -1, 0, 0);
            w.fl &= ~F_PEND_LAZY;
            if (r.pend_pos >= 0) {
                w.fl |= F_HAS_EV;
                w.ev_pos = r.pend_pos;
                w.ev_state = r.pend_state;
                w.ev_sym = r.pend_sym;
                w.ev_kind = (uint8_t) (w.tr2[r.pend_state * (T.ncls + 1) + r.pend_sym] >> 8);
                w.ev_sp = w.cur_sp;
                const bool anchored = w.cur_sp < lz_pos && lz_pos >= seg_a;
                w.ev_apos = anchored ? lz_pos : (int64_t) -1;
                w.ev_astate = anchored ? lz_s0 : 0u;
            }
        }
    };

    /* FIRST: stable stretches (sre_seg_summary_t.stable_until / stable_from), as offsets
     * from the segment start */
    constexpr uint32_t SU_UNSET = 0xffffffffu;
    uint32_t           su = SU_UNSET, run_off = 0;
    /* the (at most two) shadowed states and their rows, in uniform registers: the round
     * loop maps state <-> shadow row by comparison, not by another dependent LDS lookup */
    static_assert(SRE_SCAN_MAX_SHADOWS == 2, "two shadow rows");
    const uint32_t sh_st0 = nsh > 0 ? tabp->shadow_state[0] : 0xffffu, sh_st1 = nsh > 1 ? tabp->shadow_state[1] : 0xffffu;
    const uint32_t sh_ad0 = shadow_lds, sh_ad1 = shadow_lds + SRE_FAST_ROW_BYTES;
    const uint32_t     two = 2;                 /* shift amount of byte_x4, in a register for SDWA */
    (void) two;

    const uint32_t nrounds = WARM / TILE + G.seg_bytes / TILE;
    const uint32_t lag = (tid >> 5) & 1u;
    /* DIST = how many stages ahead the HBM loads are issued.  1: one set of staging registers,
     * a stage's loads fly while the previous round is walked.  2: two sets, taking turns — a
     * load has a whole iteration more to arrive before the wave needs it. */
    constexpr uint32_t DIST = SRE_SCAN_PREFETCH;
    __syncthreads();                        /* row tables are complete */
    auto round = [&](const uint32_t s, uint4 (&regs)[4]) __attribute__((always_inline)) {
        /* LDS operations of one wave execute in order; the fences only stop the
         * compiler from moving tile reads across the stores */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tile_store<BITS, WIDE>(regs, tile, clsx, tid, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* the next stage's HBM loads fly while this round is consumed from LDS */
        if (s + DIST <= nrounds) tile_fetch(regs, rows, tid, s + DIST);

        if (s < lag || s - lag >= nrounds) return;
        const uint32_t r = s - lag;                     /* this lane's round */
        const bool     warm_round = (r < WARM / TILE);
        if (!active || w.f(F_FINISHED) || (warm_round && !warm)) return;
        const int64_t base = seg_a - WARM + (int64_t) r * TILE;
        if (base >= seg_b || base < 0) return;
        w.anchor_pos = warm_round ? -1 : base;
        w.anchor_state = w.st;

        /* the lane's whole row of indices in one go: rows are contiguous 16-byte
         * multiples, so these wide reads are bank-conflict free */
        constexpr int ROWW = (ROWRAW + 3) / 4;                  /* dwords of indices per round */
        uint32_t      roww[ROWW < 4 ? 4 : ROWW];
        {
            const uint8_t *src = tile + tid * ROWB + (r & 1u) * ROWRAW;
            if (ROWW >= 4) {
#pragma unroll
                for (int x = 0; x < ROWW / 4; x++) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(src + 16 * x);
                    roww[4 * x] = v.x; roww[4 * x + 1] = v.y; roww[4 * x + 2] = v.z; roww[4 * x + 3] = v.w;
                }
            } else if (ROWW == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(src);
                roww[0] = v.x; roww[1] = v.y;
            } else {
                roww[0] = *reinterpret_cast<const uint32_t *>(src);
            }
        }
        /* the common round: every byte in range, no transition needs the exact
         * path, (COUNT) no match completes — one straight chain of lookups:
         * address = low half of the entry + the round's next 16-bit index (one
         * SDWA add), load, (COUNT) add the entry's count byte (one SDWA add) */
        if (base + TILE <= seg_b && !(MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT))) {
            uint32_t t = fast_lds + w.st * SRE_FAST_ROW_BYTES, cnt = 0;
            if (MODE != SRE_HIP_PIKE_COUNT && !warm_round) {
                /* a stable stretch starts (or goes on) in the state's shadow row */
                const uint32_t sha = w.st == sh_st0 ? sh_ad0 : w.st == sh_st1 ? sh_ad1 : 0u;
                if (!w.f(F_SHADOW)) {
                    if (sha) {
                        w.fl |= F_SHADOW;
                        run_off = (uint32_t) (base - seg_a);
                    } else if (su == SU_UNSET) {
                        su = (uint32_t) (base - seg_a);
                    }
                }
                if (w.f(F_SHADOW)) t = sha;
            }
#pragma unroll
            for (int j = 0; j < TILE * BITS / 8; j++) {
                uint32_t a;
                if (WIDE) {
                    a = (j & 1) ? add_w0_w<1>(t, roww[j >> 1]) : add_w0_w<0>(t, roww[j >> 1]);
                } else {
                    const uint32_t wj = roww[j >> 2];
                    const uint32_t ix = (j & 3) == 0 ? byte_x4<0>(wj, two) : (j & 3) == 1 ? byte_x4<1>(wj, two)
                                      : (j & 3) == 2 ? byte_x4<2>(wj, two) : byte_x4<3>(wj, two);
                    a = add_w0_w<2>(t, ix);
                }
                t = *(lds_u32_t) (uintptr_t) a;
                if (MODE == SRE_HIP_PIKE_COUNT) cnt = add_b3(cnt, t);
            }
            if (!(t & LDS_SLOW)) {
                /* matches completed in the round, each followed by a restart at
                 * the next byte, are only counted here (see note_span) */
                const uint32_t ridx = ((t & 0xffffu) - fast_lds) / SRE_FAST_ROW_BYTES;
                if (MODE == SRE_HIP_PIKE_COUNT) fast_span_done(base, TILE, w.st, cnt, (t & LDS_FRESH) != 0, warm_round);
                w.st = MODE == SRE_HIP_PIKE_COUNT ? ridx : ridx <= nst ? ridx : ridx == nst + 1 ? sh_st0 : sh_st1;
                if (MODE != SRE_HIP_PIKE_COUNT && w.f(F_SHADOW) && ridx <= nst) {
                    /* left the shadow rows: the stable stretch ended in this round */
                    w.fl &= ~F_SHADOW;
                    if (su == SU_UNSET) su = (uint32_t) (base - seg_a);
                }
                if (r + 1 == WARM / TILE) {
                    s_in = w.st | (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT) ? SRE_STATE_SKIP : 0u);
                    settle();
                    w.cur_sp = -1;
                    if (w.f(F_HAS_EV)) w.ev_sp = -1;
                    if (w.st == 0) w.fl |= F_FINISHED;
                    if (MODE == SRE_HIP_PIKE_COUNT) {
                        w.set(F_IN_PENDING, w.f(F_HAS_EV));
                        in_pe_pos = w.ev_pos;
                        in_pe_state = w.ev_state;
                        in_pe_sym = w.ev_sym;
                    }
                }
                return;
            }
        }
        if (MODE != SRE_HIP_PIKE_COUNT && !warm_round) {
            /* a round that takes the exact path ends any stable stretch */
            w.fl &= ~F_SHADOW;
            if (su == SU_UNSET) su = (uint32_t) (base - seg_a);
        }
        /* otherwise group by group (16 bytes); a real loop, the indices of a
         * group re-read from the tile: this is the uncommon path and must not
         * weigh on the registers of the common one */
#pragma unroll 1
        for (uint32_t q = 0; q < TILE / 16; q++) {
            const int64_t gp = base + q * 16;
            if (gp >= seg_b || w.f(F_FINISHED)) break;
            const int64_t g_end = gp + 16 <= seg_b ? gp + 16 : seg_b;   /* ragged tail of the stream */
            bool          exact = (g_end != gp + 16) || (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT));
            uint32_t      t = fast_lds + w.st * SRE_FAST_ROW_BYTES, cnt = 0;
            if (!exact) {
                constexpr int GW = GIDX * (WIDE ? 2 : 1) / 4;           /* dwords of indices per group */
                uint32_t      gw[GW];
                const uint32_t *gsrc = reinterpret_cast<const uint32_t *>(tile + tid * ROWB + (r & 1u) * ROWRAW) + q * GW;
#pragma unroll
                for (int x = 0; x < GW; x++) gw[x] = gsrc[x];
#pragma unroll
                for (int j = 0; j < GIDX; j++) {
                    const uint32_t ix = WIDE ? (gw[j >> 1] >> ((j & 1) * 16)) & 0xffffu
                                             : ((gw[j >> 2] >> ((j & 3) * 8)) & 0xffu) << 2;
                    t = *(lds_u32_t) (uintptr_t) ((t & 0xffffu) + ix);
                    if (MODE == SRE_HIP_PIKE_COUNT) cnt += t >> LDS_CNT_SHIFT;
                }
                exact = (t & LDS_SLOW) != 0;        /* the trap row keeps the flag */
            }
            if (exact) {
                settle();
                slow_run<MODE>(w, gp, g_end, warm_round, seed);
            } else {
                const uint32_t st1 = ((t & 0xffffu) - fast_lds) / SRE_FAST_ROW_BYTES;    /* ordinary rows only on this path */
                if (MODE == SRE_HIP_PIKE_COUNT) fast_span_done(gp, 16, w.st, cnt, (t & LDS_FRESH) != 0, warm_round);
                w.st = st1;
            }
        }
        if (r + 1 == WARM / TILE) {
            /* end of the warm-up: what this lane assumes about its entry */
            s_in = w.st | (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT) ? SRE_STATE_SKIP : 0u);
            settle();
            w.cur_sp = -1;                  /* search starts seen in the warm-up are not verified */
            if (w.f(F_HAS_EV)) w.ev_sp = -1;
            if (w.st == 0) w.fl |= F_FINISHED;
            if (MODE == SRE_HIP_PIKE_COUNT) {
                w.set(F_IN_PENDING, w.f(F_HAS_EV));      /* ... nor is the pending match: the chain check compares it */
                in_pe_pos = w.ev_pos;
                in_pe_state = w.ev_state;
                in_pe_sym = w.ev_sym;
            }
        }
    
    };
    uint4 regs[4];
    tile_fetch(regs, rows, tid, 0);
    if (DIST == 1) {
        for (uint32_t s = 0; s <= nrounds; s++) round(s, regs);
    } else {
        uint4 regs2[4];
        if (nrounds >= 1) tile_fetch(regs2, rows, tid, 1);
        for (uint32_t s = 0; s <= nrounds; s += 2) {
            round(s, regs);
            if (s + 1 <= nrounds) round(s + 1, regs2);
        }
    }

    if (!active) return;

    /* the lane that owns the end of the stream performs the EOF step(s) — unless more
     * chunks of the stream follow */
    if (last_seg && !w.f(F_FINISHED) && !(G.flags & SRE_GEOM_NO_EOF)) {
        settle();
        w.anchor_pos = -1;
        slow_run<MODE>(w, w.n, w.n + 1, false, 0);
    }
    /* the segment's last completed match may sit in a pure-fast span: recover it */
    settle();

    sre_seg_summary_t out;
    {
        const uint32_t seg_len = (uint32_t) (seg_b - seg_a);
        if (MODE == SRE_HIP_PIKE_COUNT || (w.f(F_FINISHED) && su == SU_UNSET)) su = 0;
        out.stable_until = su == SU_UNSET ? seg_len : su;
        out.stable_from = (MODE != SRE_HIP_PIKE_COUNT && w.f(F_SHADOW)) ? run_off : seg_len;
    }
    out.s_in = w.f(F_UNRESOLVED) ? 0xffffffffu : s_in;
    out.s_out = w.st | (MODE == SRE_HIP_PIKE_COUNT && w.f(F_SKIP_NEXT) ? SRE_STATE_SKIP : 0u);
    out.flags = 0;
    out.pad = 0;
    out.count = w.count;
    out.term_pos = w.term_pos;
    out.cur_sp = w.cur_sp;
    if (w.f(F_FINISHED) && w.term_pos >= 0) out.flags |= SRE_SUM_TERM;
    if (w.f(F_ERROR)) out.flags |= SRE_SUM_ERROR;
    if (w.f(F_HAS_EV)) out.flags |= SRE_SUM_PENDING;
    if (w.f(F_LM_VALID)) out.flags |= SRE_SUM_LASTEV;
    if (MODE == SRE_HIP_PIKE_COUNT && w.f(F_IN_PENDING)) out.flags |= SRE_SUM_IN_PENDING;
    if (MODE != SRE_HIP_PIKE_COUNT && out.stable_until == G.seg_bytes && !last_seg) out.flags |= SRE_SUM_STABLE;
    out.in_pe_pos = in_pe_pos;
    out.in_pe_state = in_pe_state;
    out.in_pe_sym = in_pe_sym;
    out.pe_state = w.ev_state;
    out.pe_sym = w.ev_sym;
    out.pe_pos = w.ev_pos;
    out.pe_sp = w.ev_sp;
    out.lm_state = w.lm_state;
    out.lm_sym = w.lm_sym;
    out.lm_pos = w.lm_pos;
    out.lm_sp = w.lm_sp;
    out.lm_apos = w.lm_apos;
    out.lm_astate = w.lm_astate;
    out.pe_apos = w.ev_apos;
    out.pe_astate = w.ev_astate;
    sum[g] = out;
    if (MODE != SRE_HIP_PIKE_COUNT && G.digest != nullptr) {
        sre_seg_digest_t dg;
        dg.s_in = out.s_in;
        dg.s_out = out.s_out;
        dg.bits = ((out.flags & SRE_SUM_TERM) ? 2u : 0u) | ((out.flags & SRE_SUM_LASTEV) ? 4u : 0u) | (out.cur_sp >= 0 ? 8u : 0u)
                  | ((out.flags & SRE_SUM_STABLE) ? 0u : 16u);
        dg.count = (uint32_t) out.count;
        G.digest[g] = dg;
    }
}

/* ============================================================ exact entry states */

/*
 * A stream whose automaton never "forgets" (a state that rotates with the input:
 * x(?:[^y]{3})*y behind an x) defeats speculation: a fix-up round is only sure to
 * repair ONE segment, so the rounds would go on for as many segments as the stream
 * has.  After two rounds that did not settle a FIRST / Thompson scan, the entry
 * state of every remaining segment is computed exactly instead:
 *   sre_k_seg_functions  one WAVE per segment, lane l = entry state l: all 64 lanes
 *                        read the same bytes and walk the LDS fast table from their
 *                        own state — the segment's whole transition function
 *                        (64 bytes) at the cost of one lane's walk;
 *   sre_k_fn_chunks      composes 256 consecutive functions (again lane = state);
 *   sre_k_fn_resolve     per stream: from the verified prefix's exit state through
 *                        partial chunks and chunk compositions;
 *   sre_k_fn_fill        per chunk: the entry state of each of its segments.
 * One more scan pass with these entry states is exact in every lane.  (COUNT keeps
 * the speculative rounds: where its searches restart depends on match ends, which
 * is not a function of the state alone.)
 */
#define SRE_FN_CHUNK 256u

template <int BITS>
__global__ __launch_bounds__(256) void
sre_k_seg_functions(const sre_scan_tables_t *__restrict__ tabp, sre_scan_geom_t G,
                    const sre_stream_status_t *__restrict__ status, uint8_t *__restrict__ fn)
{
    constexpr int STRIDE = 8 / BITS;
    constexpr int GIDX = 16 / STRIDE;
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint16_t clsx[BITS == 8 ? 1 : 8 / BITS][256];
    const uint32_t tid = threadIdx.x, nst = tabp->nstates, nsym = tabp->ncls + 1;
    /* LDS: [fast rows + trap][class map][transitions, 2 B each] */
    uint32_t *fast = reinterpret_cast<uint32_t *>(lds);
    uint8_t  *clsl = lds + (nst + 1) * SRE_FAST_ROW_BYTES;
    uint16_t *tr2 = reinterpret_cast<uint16_t *>(clsl + 256);
    constexpr uint32_t LDS_SLOW = 1u << 16;
    const uint32_t fast_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) lds;
    const uint32_t trap_lds = fast_lds + nst * SRE_FAST_ROW_BYTES;
    for (uint32_t i = tid; i < nst * 256; i += 256) {
        const uint32_t gl = tabp->fast[i];
        fast[i] = (gl & SRE_FAST_SLOW) ? (trap_lds | LDS_SLOW) : fast_lds + (gl & ~(SRE_FAST_ROW_BYTES - 1));
    }
    fast[nst * 256 + tid] = trap_lds | LDS_SLOW;
    clsl[tid] = tabp->cls[tid];
    for (uint32_t i = tid; i < nst * nsym; i += 256) tr2[i] = tabp->trans2[i];
    if (BITS != 8) {
#pragma unroll
        for (int u = 0; u < 8 / BITS; u++) clsx[u][tid] = (uint16_t) ((uint32_t) tabp->cls[tid] << (BITS * u + 2));
    }
    __syncthreads();

    const uint32_t lane = tid & 63u;
    const uint64_t g = (uint64_t) blockIdx.x * 4 + (tid >> 6);
    if (g >= G.nsegs) return;
    const uint32_t s = stream_of(G, g);
    const uint64_t k = g - geom_first(G, s);
    if (status[s].done || (int64_t) k < status[s].first_bad) return;
    if (lane >= nst) {
        fn[g * 64 + lane] = 0;
        return;
    }
    const uint8_t *data = geom_ptr(G, s);
    const int64_t  n = (int64_t) geom_len(G, s);
    int64_t        p = (int64_t) k * G.seg_bytes, seg_b = p + G.seg_bytes;
    if (seg_b > n) seg_b = n;
    uint32_t cur = lane;
    const bool count_mode = tabp->mode == SRE_HIP_PIKE_COUNT;
    auto exact = [&](int64_t from, int64_t to) {
        if (!count_mode) {
            for (int64_t q = from; q < to; q++) cur = tr2[cur * nsym + clsl[data[q]]] & 0xffu;
            return;
        }
        /* COUNT: the caller's restarts belong to the function (slow_run): a search that ends with a match is
         * followed by one from the match end (a byte further after an empty match), from the list the byte in
         * front selects.  A match recorded in front of this span is taken to end right here (what a FRESH
         * state says; elsewhere an approximation — the chain check decides, a wrong entry state costs a round). */
        int64_t  pend = -1;
        uint32_t pkind = 0, guard = 0;
        for (int64_t q = from; q < to; q++) {
            const uint32_t sym = clsl[data[q]];
            const uint32_t t2 = tr2[cur * nsym + sym];
            if (t2 >> 8) {
                pend = q;
                pkind = t2 >> 8;
            }
            cur = t2 & 0xffu;
            if (cur != 0) continue;
            int64_t sp = q + 1;
            if (pend >= 0) {
                const bool    pop = pkind == EV_POP || pkind == SRE_DEV_EV_POP_FULL;
                const bool    empty = pkind == EV_POP || pkind == SRE_DEV_EV_DONE_EMPTY;
                const int64_t e = pop ? pend : pend + 1;
                sp = empty ? e + 1 : e;
            } else if (++guard < 2) {
                sp = q;                         /* the match in flight ended in front of this byte: read it again */
            }
            if (pend >= 0) guard = 0;
            if (sp < 1) sp = 1;
            if (sp > n) sp = n;
            cur = tabp->init[restart_variant_of(*tabp, data[sp - 1])];
            pend = -1;
            q = sp - 1;
        }
    };
    for (; p + 16 <= seg_b; p += 16) {
        const sre_u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) sre_u32x4_unaligned *>(
            reinterpret_cast<uintptr_t>(data + p));
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
        uint32_t       t = fast_lds + cur * SRE_FAST_ROW_BYTES;
#pragma unroll
        for (int j = 0; j < GIDX; j++) {
            uint32_t ix = 0;
#pragma unroll
            for (int u = 0; u < STRIDE; u++) {
                const int      b = j * STRIDE + u;
                const uint32_t c = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                ix |= BITS == 8 ? c << 2 : (uint32_t) clsx[u][c];
            }
            t = *(lds_u32_t) (uintptr_t) ((t & 0xffffu) + ix);
        }
        if (t & LDS_SLOW) exact(p, p + 16);
        else cur = ((t & 0xffffu) - fast_lds) / SRE_FAST_ROW_BYTES;
    }
    exact(p, seg_b);
    fn[g * 64 + lane] = (uint8_t) cur;
}

__global__ __launch_bounds__(64) void
sre_k_fn_chunks(uint64_t nsegs, const uint8_t *__restrict__ fn, uint8_t *__restrict__ comp)
{
    const uint64_t c = blockIdx.x;
    uint32_t       v = threadIdx.x;
    const uint64_t g0 = c * SRE_FN_CHUNK, g1 = g0 + SRE_FN_CHUNK <= nsegs ? g0 + SRE_FN_CHUNK : nsegs;
    for (uint64_t g = g0; g < g1; g++) v = fn[g * 64 + (v & 63u)];
    comp[c * 64 + threadIdx.x] = (uint8_t) v;
}

__global__ void
sre_k_fn_resolve(sre_scan_geom_t G, const sre_seg_summary_t *__restrict__ sum,
                 const sre_stream_status_t *__restrict__ status, const uint8_t *__restrict__ fn,
                 const uint8_t *__restrict__ comp, uint8_t *__restrict__ entry, uint8_t *__restrict__ chunk_entry)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.nstreams || status[s].done || status[s].first_bad < 1) return;
    const uint64_t first = geom_first(G, s), g_hi = geom_first(G, s + 1);
    uint64_t       g = first + (uint64_t) status[s].first_bad;
    uint32_t       v = sum[g - 1].s_out & 63u;          /* the verified prefix's exit state */
    while (g < g_hi) {
        if (g % SRE_FN_CHUNK == 0 && g + SRE_FN_CHUNK <= g_hi) {
            chunk_entry[g / SRE_FN_CHUNK] = (uint8_t) v;
            v = comp[(g / SRE_FN_CHUNK) * 64 + v];
            g += SRE_FN_CHUNK;
        } else {
            entry[g] = (uint8_t) v;
            v = fn[g * 64 + v];
            g++;
        }
    }
}

__global__ void
sre_k_fn_fill(uint64_t nsegs, const uint8_t *__restrict__ fn, const uint8_t *__restrict__ chunk_entry,
              uint8_t *__restrict__ entry)
{
    const uint64_t c = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (c * SRE_FN_CHUNK >= nsegs || chunk_entry[c] == 0xffu) return;
    uint32_t v = chunk_entry[c];
    for (uint64_t g = c * SRE_FN_CHUNK; g < (c + 1) * SRE_FN_CHUNK && g < nsegs; g++) {
        entry[g] = (uint8_t) v;
        v = fn[g * 64 + (v & 63u)];
    }
}

/* ===================================================================== verify */

/*
 * Chain check and reduction, grid-wide (one lane per segment), in three small
 * kernels so that the single-stream case (hundreds of thousands of segments)
 * is not serialised on one workgroup:
 *   A  first broken link / first segment that ended the scan   (atomicMin)
 *   B  over the verified-and-needed prefix: match count, last match's segment
 *   C  per stream: assemble the status word (+ the search start of the match)
 */
struct VerifyAcc {
    unsigned long long bad, end;        /* init ~0 */
    unsigned long long count, evseg, spseg;     /* init 0 */
    unsigned long long unst;                    /* init 0: 1 + last segment in front of evseg that is not stable */
    unsigned long long unst_end;                /* init 0: 1 + last segment of the verified prefix that is not stable */
};

__global__ __launch_bounds__(256) void
sre_k_verify_a(sre_scan_geom_t G, const sre_seg_summary_t *__restrict__ sum, VerifyAcc *__restrict__ acc,
               int mode)
{
    SRE_TAIL_PRIO();
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool     valid = g < G.nsegs;
    const uint32_t s = stream_of(G, valid ? g : G.nsegs - 1);
    unsigned long long kb = ~0ull, ke = ~0ull;      /* this lane's candidates for acc[s].bad / .end */
    if (valid) {
        const uint64_t k = g - geom_first(G, s);
        const uint32_t s_in = sum[g].s_in;
        bool           bad = s_in == 0xffffffffu;
        if (k > 0) {
            const sre_seg_summary_t &p = sum[g - 1], &c = sum[g];
            if (s_in != p.s_out) bad = true;
            if (mode == SRE_HIP_PIKE_COUNT) {
                /* the same entry state can hold different pending matches */
                const bool pp = (p.flags & SRE_SUM_PENDING) != 0, cp = (c.flags & SRE_SUM_IN_PENDING) != 0;
                if (pp != cp) bad = true;
                if (pp && cp && (p.pe_pos != c.in_pe_pos || p.pe_state != c.in_pe_state || p.pe_sym != c.in_pe_sym)) {
                    bad = true;
                }
            }
        }
        if (bad) kb = k;
        if (sum[g].flags & SRE_SUM_TERM) ke = k;
    }
    /* One atomic per wave at most, and none when the word already holds less: with a match in reach of
     * every segment (`foo` over "foo foo ..", first match) each of the 262 144 lanes of a 64 MiB stream
     * reported TERM with an atomicMin on ONE address — 3 ms for a scan of 0.13 (profiles/r03_floor_probe.json).
     * (A stale read is only ever larger than the word: the test errs towards the atomic.) */
    const uint32_t s0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) s);
    if (__builtin_amdgcn_ballot_w64(s != s0) == 0) {
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long ob = __shfl_xor(kb, d, 64), oe = __shfl_xor(ke, d, 64);
            kb = ob < kb ? ob : kb;
            ke = oe < ke ? oe : ke;
        }
        if ((threadIdx.x & 63u) != 0) kb = ke = ~0ull;
    }
    if (kb != ~0ull && kb < __hip_atomic_load(&acc[s].bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&acc[s].bad, kb);
    if (ke != ~0ull && ke < __hip_atomic_load(&acc[s].end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&acc[s].end, ke);
}

/* (256 threads cover SRE_VERIFY_SPAN = 1024 segments: one global atomic per 1024 segments
 * and stream — they all land on one address, a 4 GiB stream is 258K segments — from a
 * workgroup small enough for a spare slot while the next scan holds the chip: as 1024-thread
 * workgroups these kernels waited ~0.45 ms for a CU to drain, profiles/r02_kernel_stats.csv) */
#define SRE_VERIFY_SPAN 1024u

__global__ __launch_bounds__(256) void
sre_k_verify_b(sre_scan_geom_t G, const sre_seg_summary_t *__restrict__ sum, VerifyAcc *__restrict__ acc)
{
    SRE_TAIL_PRIO();
    __shared__ unsigned long long sh_count, sh_ev, sh_sp;
    const uint64_t g0 = (uint64_t) blockIdx.x * SRE_VERIFY_SPAN;
    const uint64_t glast = (g0 + SRE_VERIFY_SPAN - 1 < G.nsegs) ? g0 + SRE_VERIFY_SPAN - 1 : G.nsegs - 1;
    const uint32_t s_first = stream_of(G, g0), s_last = stream_of(G, glast);
    const bool     uniform = (s_first == s_last);       /* whole span inside one stream */
    if (threadIdx.x == 0) {
        sh_count = 0;
        sh_ev = 0;
        sh_sp = 0;
    }
    __syncthreads();
    /* whole span inside one stream (the usual case): reduce in the wave first — a
     * shared-memory atomic per lane serialises 256 ways when every segment counts */
    unsigned long long my_count = 0, my_ev = 0, my_sp = 0;
    for (uint32_t it = 0; it < SRE_VERIFY_SPAN / 256; it++) {
        const uint64_t g = g0 + it * 256u + threadIdx.x;
        if (g >= G.nsegs) break;
        const uint32_t s = uniform ? s_first : stream_of(G, g);
        const uint64_t k = g - geom_first(G, s);
        const uint64_t nseg = geom_first(G, s + 1) - geom_first(G, s);
        uint64_t       bad = acc[s].bad, end = acc[s].end;
        if (bad > nseg) bad = nseg;
        if (end > nseg) end = nseg;
        const uint64_t limit = end < bad ? end + 1 : bad;
        if (k < limit) {
            const sre_seg_summary_t &c = sum[g];
            if (uniform) {
                my_count += (unsigned long long) c.count;
                if (c.flags & SRE_SUM_LASTEV) my_ev = (unsigned long long) k + 1;
                if (c.cur_sp >= 0) my_sp = (unsigned long long) k + 1;     /* latest segment at whose end a search start is known */
            } else {
                if (c.count) atomicAdd(&acc[s].count, (unsigned long long) c.count);
                if (c.flags & SRE_SUM_LASTEV) atomicMax(&acc[s].evseg, (unsigned long long) k + 1);
                if (c.cur_sp >= 0) atomicMax(&acc[s].spseg, (unsigned long long) k + 1);
            }
        }
    }
    if (uniform) {
        for (int d = 32; d >= 1; d >>= 1) {
            my_count += __shfl_down(my_count, d, 64);
            const unsigned long long e2 = __shfl_down(my_ev, d, 64), s2 = __shfl_down(my_sp, d, 64);
            my_ev = e2 > my_ev ? e2 : my_ev;
            my_sp = s2 > my_sp ? s2 : my_sp;
        }
        if ((threadIdx.x & 63u) == 0) {
            if (my_count) atomicAdd(&sh_count, my_count);
            if (my_ev) atomicMax(&sh_ev, my_ev);
            if (my_sp) atomicMax(&sh_sp, my_sp);
        }
    }
    __syncthreads();
    if (uniform && threadIdx.x == 0) {
        if (sh_count) atomicAdd(&acc[s_first].count, sh_count);
        if (sh_ev) atomicMax(&acc[s_first].evseg, sh_ev);
        if (sh_sp) atomicMax(&acc[s_first].spseg, sh_sp);
    }
}

/* FIRST: the last segment in front of the event's that is not SRE_SUM_STABLE — between
 * the two the automaton sat in one state whose neutral threads looped in place, and the
 * capture walker crosses all of them in one jump */
__global__ __launch_bounds__(256) void
sre_k_verify_b2(sre_scan_geom_t G, const sre_seg_summary_t *__restrict__ sum, VerifyAcc *__restrict__ acc)
{
    SRE_TAIL_PRIO();
    __shared__ unsigned long long sh_max, sh_max_end;
    const uint64_t g0 = (uint64_t) blockIdx.x * SRE_VERIFY_SPAN;
    const uint64_t glast = (g0 + SRE_VERIFY_SPAN - 1 < G.nsegs) ? g0 + SRE_VERIFY_SPAN - 1 : G.nsegs - 1;
    const uint32_t s_first = stream_of(G, g0), s_last = stream_of(G, glast);
    const bool     uniform = (s_first == s_last);       /* whole span inside one stream */
    if (threadIdx.x == 0) sh_max = sh_max_end = 0;
    __syncthreads();
    unsigned long long mine = 0, mine_end = 0;
    for (uint32_t it = 0; it < SRE_VERIFY_SPAN / 256; it++) {
        const uint64_t g = g0 + it * 256u + threadIdx.x;
        if (g >= G.nsegs) break;
        const uint32_t s = uniform ? s_first : stream_of(G, g);
        const uint64_t k = g - geom_first(G, s);
        const uint64_t evseg = acc[s].evseg;        /* 1 + the event's segment, 0 none */
        const uint64_t nseg = geom_first(G, s + 1) - geom_first(G, s);
        if (!(sum[g].flags & SRE_SUM_STABLE)) {
            unsigned long long m1 = 0, m2 = 0;
            if (evseg != 0 && k + 1 < evseg) m1 = k + 1;
            /* (streaming: the walks that start at the end of a chunk; its last segment is
             * never flagged stable and is crossed by its own stable prefix / suffix) */
            if (k + 1 < nseg) m2 = k + 1;
            if (uniform) {
                mine = m1 > mine ? m1 : mine;
                mine_end = m2 > mine_end ? m2 : mine_end;
            } else {
                if (m1) atomicMax(&acc[s].unst, m1);
                if (m2) atomicMax(&acc[s].unst_end, m2);
            }
        }
    }
    if (uniform) {
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_down(mine, d, 64), o2 = __shfl_down(mine_end, d, 64);
            mine = o > mine ? o : mine;
            mine_end = o2 > mine_end ? o2 : mine_end;
        }
        if ((threadIdx.x & 63u) == 0) {
            if (mine) atomicMax(&sh_max, mine);
            if (mine_end) atomicMax(&sh_max_end, mine_end);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (sh_max) atomicMax(&acc[s_first].unst, sh_max);
            if (sh_max_end) atomicMax(&acc[s_first].unst_end, sh_max_end);
        }
    }
}

/* the status word of one stream from its accumulated chain-check results; `sum` = the
 * stream's first summary */
#define SRE_EV_SP_SEARCH ((int64_t) -2)
__device__ inline sre_stream_status_t
assemble_status(const sre_scan_tables_t &T, const sre_seg_summary_t *__restrict__ sum, uint64_t nseg, const VerifyAcc &acc,
                bool defer_search = false)
{
    uint64_t       bad = acc.bad, end = acc.end;
    if (bad > nseg) bad = nseg;
    if (end > nseg) end = nseg;
    const bool     done = (end < bad) || (bad >= nseg);
    const uint64_t limit = end < bad ? end + 1 : bad;
    const uint64_t evseg = acc.evseg;

    sre_stream_status_t st;
    st.first_bad = (int64_t) bad;
    st.limit = (int64_t) limit;
    st.done = done ? 1 : 0;
    st.error = 0;
    st.need_maps = 0;
    st.pad = 0;
    st.count = (int64_t) acc.count;
    st.rc = RC_DECLINED;
    st.ev_pos = st.ev_sp = -1;
    st.ev_state = st.ev_sym = 0;
    st.ev_apos = -1;
    st.ev_astate = 0;
    st.valid_from = 0;
    st.ev_seg = -1;
    st.unst_seg = acc.unst ? (int64_t) acc.unst - 1 : -1;
    st.unst_end = acc.unst_end ? (int64_t) acc.unst_end - 1 : -1;
    if (done && evseg > 0) {
        const sre_seg_summary_t &c = sum[evseg - 1];
        st.ev_apos = c.lm_apos;
        st.ev_astate = c.lm_astate;
        st.ev_state = c.lm_state;
        st.ev_sym = c.lm_sym;
        st.ev_pos = c.lm_pos;
        st.ev_sp = c.lm_sp;
        st.ev_seg = (int64_t) evseg - 1;
        st.rc = T.trans[(size_t) c.lm_state * (T.ncls + 1) + c.lm_sym].regex;
        if (T.mode != SRE_HIP_PIKE_COUNT) st.count = 1;
        if (st.ev_sp < 0) {
            /* the match's search began in an earlier segment: the latest start
             * known before it.  (A start known at or after the match's own
             * segment belongs to a later search.) */
            int64_t        sp = 0;
            const uint64_t spseg = acc.spseg;
            if (T.mode != SRE_HIP_PIKE_COUNT || spseg == 0) {
                sp = 0;                     /* one search per stream, from its start */
            } else if (spseg < evseg) {
                sp = sum[spseg - 1].cur_sp;
            } else if (defer_search) {
                sp = SRE_EV_SP_SEARCH;      /* the caller looks for it with the whole wave (sre_k_verify_c) */
            } else {
                for (int64_t q = (int64_t) evseg - 2; q >= 0; q--) {
                    if (sum[q].cur_sp >= 0) {
                        sp = sum[q].cur_sp;
                        break;
                    }
                }
            }
            st.ev_sp = sp;
        }
    }
    if (done && end < nseg && (sum[end].flags & SRE_SUM_ERROR)) st.error = 1;
    return st;
}

__global__ __launch_bounds__(64) void
sre_k_verify_c(sre_scan_tables_t T, sre_scan_geom_t G, const sre_seg_summary_t *__restrict__ sum,
               VerifyAcc *__restrict__ accs, sre_stream_status_t *__restrict__ status)
{
    SRE_TAIL_PRIO();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool     valid = s < G.nstreams;
    const uint32_t lane = threadIdx.x & 63u;
    sre_stream_status_t st;
    uint64_t            first = 0;
    st.ev_sp = -1;
    st.ev_seg = -1;
    if (valid) {
        /* take this stream's accumulator and leave it reset for the next pass */
        const VerifyAcc acc = accs[s];
        accs[s].bad = accs[s].end = ~0ull;
        accs[s].count = accs[s].evseg = accs[s].spseg = accs[s].unst = accs[s].unst_end = 0;
        first = geom_first(G, s);
        const uint64_t nseg = geom_first(G, s + 1) - first;
        st = assemble_status(T, sum + first, nseg, acc, true);
    }
    /* COUNT: the search of the last match began in an earlier segment and a later one knows a start too
     * (`a+` over a stream of a: the match spans it, the search behind it starts at its end): the latest
     * start known in front of the match's segment, looked for by the whole wave, 256 segments per trip —
     * one lane walking the summaries backwards took 36 ms over the 131 072 segments of 64 MiB. */
    uint64_t need = __builtin_amdgcn_ballot_w64(valid && st.ev_sp == SRE_EV_SP_SEARCH);
    while (need) {
        const int      l = __builtin_ctzll(need);
        need &= need - 1;
        const uint64_t f = __shfl(first, l, 64);
        const int64_t  top = __shfl(st.ev_seg, l, 64) - 1;     /* segments [0, top] are in front of the match's */
        int64_t        sp = 0;
        bool           found = false;
        for (int64_t hi = top; hi >= 0 && !found; hi -= 256) {
            int64_t v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t q = hi - (int64_t) lane - 64 * j;
                v[j] = q >= 0 ? sum[f + (uint64_t) q].cur_sp : (int64_t) -1;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t b = __builtin_amdgcn_ballot_w64(v[j] >= 0);
                if (!found && b) {
                    sp = __shfl(v[j], __builtin_ctzll(b), 64);      /* the lowest lane holds the highest segment */
                    found = true;
                }
            }
        }
        if ((int) lane == l) st.ev_sp = sp;
    }
    if (valid) status[s] = st;
}

/*
 * The chain check of ONE stream (FIRST / Thompson) by one workgroup — phases A, B, B2 and C
 * of the kernels above between workgroup barriers instead of kernel boundaries: a chunk of
 * a chunked stream has a few thousand segments, and four dependent launches cost more
 * than the scan itself there.  Every thread of the workgroup calls it; the status is in
 * *out (shared memory) behind the last barrier, and thread 0 has stored it to *status.
 */
#define SRE_VERIFY_ONE_MAX SRE_VERIFY_ONE_SEGS     /* segments one workgroup checks (8 per thread) */

template <int NT>
__device__ void
verify_one_stream(const sre_scan_tables_t &T, const sre_seg_summary_t *__restrict__ sum, uint64_t nseg,
                  sre_stream_status_t *__restrict__ status, VerifyAcc *sh_acc, sre_stream_status_t *out,
                  const sre_seg_digest_t *__restrict__ dig = nullptr)
{
    constexpr int  PER = SRE_VERIFY_ONE_MAX / NT;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        sh_acc->bad = sh_acc->end = ~0ull;
        sh_acc->count = sh_acc->evseg = sh_acc->spseg = sh_acc->unst = sh_acc->unst_end = 0;
    }
    /* every summary is read ONCE, all of a thread's loads in flight together; the phases
     * below work on what it kept: [0] link broken [1] TERM [2] LASTEV [3] search start known
     * [4] not stable, and the segment's match count */
    uint32_t bits[PER], cnt[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint64_t k = (uint64_t) i * NT + tid;
        bits[i] = 0;
        cnt[i] = 0;
        if (k < nseg && dig != nullptr) {
            /* (the scan kernel's digest: 16 bytes a segment, coalesced) */
            const sre_seg_digest_t c = dig[k];
            const uint32_t         prev_out = k > 0 ? dig[k - 1].s_out : c.s_in;
            bits[i] = ((c.s_in == 0xffffffffu || c.s_in != prev_out) ? 1u : 0u) | c.bits;
            cnt[i] = c.count;
        } else if (k < nseg) {
            const sre_seg_summary_t &c = sum[k];
            const uint32_t s_in = c.s_in, fl = c.flags;
            const uint32_t prev_out = k > 0 ? sum[k - 1].s_out : s_in;
            bits[i] = ((s_in == 0xffffffffu || s_in != prev_out) ? 1u : 0u) | ((fl & SRE_SUM_TERM) ? 2u : 0u)
                      | ((fl & SRE_SUM_LASTEV) ? 4u : 0u) | (c.cur_sp >= 0 ? 8u : 0u) | ((fl & SRE_SUM_STABLE) ? 0u : 16u);
            cnt[i] = (uint32_t) c.count;
        }
    }
    __syncthreads();
    /* A: first broken link, first segment that ended the search */
    {
        unsigned long long bad = ~0ull, end = ~0ull;
#pragma unroll
        for (int i = PER - 1; i >= 0; i--) {
            const uint64_t k = (uint64_t) i * NT + tid;
            if (bits[i] & 1u) bad = k;
            if (bits[i] & 2u) end = k;
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long b2 = __shfl_down(bad, d, 64), e2 = __shfl_down(end, d, 64);
            bad = b2 < bad ? b2 : bad;
            end = e2 < end ? e2 : end;
        }
        if ((tid & 63u) == 0) {
            if (bad != ~0ull) atomicMin(&sh_acc->bad, bad);
            if (end != ~0ull) atomicMin(&sh_acc->end, end);
        }
    }
    __syncthreads();
    /* B: over the verified-and-needed prefix */
    {
        uint64_t bad = sh_acc->bad, end = sh_acc->end;
        if (bad > nseg) bad = nseg;
        if (end > nseg) end = nseg;
        const uint64_t     limit = end < bad ? end + 1 : bad;
        unsigned long long c = 0, ev = 0, sp = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint64_t k = (uint64_t) i * NT + tid;
            if (k < limit) {
                c += cnt[i];
                if (bits[i] & 4u) ev = k + 1;
                if (bits[i] & 8u) sp = k + 1;
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            c += __shfl_down(c, d, 64);
            const unsigned long long e2 = __shfl_down(ev, d, 64), s2 = __shfl_down(sp, d, 64);
            ev = e2 > ev ? e2 : ev;
            sp = s2 > sp ? s2 : sp;
        }
        if ((tid & 63u) == 0) {
            if (c) atomicAdd(&sh_acc->count, c);
            if (ev) atomicMax(&sh_acc->evseg, ev);
            if (sp) atomicMax(&sh_acc->spseg, sp);
        }
    }
    __syncthreads();
    /* B2: the last segments that are not stable (sre_k_verify_b2) */
    {
        const uint64_t     evseg = sh_acc->evseg;
        unsigned long long mine = 0, mine_end = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint64_t k = (uint64_t) i * NT + tid;
            if (k < nseg && (bits[i] & 16u)) {
                if (evseg != 0 && k + 1 < evseg) mine = k + 1;
                if (k + 1 < nseg) mine_end = k + 1;
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_down(mine, d, 64), o2 = __shfl_down(mine_end, d, 64);
            mine = o > mine ? o : mine;
            mine_end = o2 > mine_end ? o2 : mine_end;
        }
        if ((tid & 63u) == 0) {
            if (mine) atomicMax(&sh_acc->unst, mine);
            if (mine_end) atomicMax(&sh_acc->unst_end, mine_end);
        }
    }
    __syncthreads();
    if (tid == 0) {
        *out = assemble_status(T, sum, nseg, *sh_acc);
        *status = *out;
    }
    __syncthreads();
}

/* =================================================================== captures */

/* First segment whose recorded entry state belongs to the search that started
 * at sp and holds the final event (Tracer::entry_state): the one behind the
 * segment whose lane (re)started that search, i.e. the first lane from sp's
 * segment on that ends with cur_sp == sp.  One search per stream (FIRST /
 * Thompson): the lane of sp's segment itself. */
__device__ inline int64_t
first_valid_segment(const sre_seg_summary_t *sum, int64_t sp, int64_t seg_bytes, int64_t ev_seg)
{
    for (int64_t k = sp / seg_bytes; k <= ev_seg; k++) {
        if (sum[k].cur_sp == sp) return k + 1;
    }
    return ev_seg + 1;          /* nothing recorded can be taken: replay from sp */
}

/*
 * One lane per stream.  The winning thread's capture vector is rebuilt by
 * walking its lineage backwards from the match event (sre_dfa.h).  The state
 * before a position is recomputed on demand, one segment-sized block at a
 * time, forwards from that segment's verified entry state (2 B of scratch per
 * byte); then parent links are followed backwards, collecting the SAVE slots
 * still unresolved.  The walk ends at the ".*?" thread (nothing saved yet), at
 * a thread re-seeded by the leading-byte skip, at the start of the search, or
 * when every slot is known.
 */
/* (every method is force-inlined: the walker is one lane, and a Tracer whose address escapes into
 * a call lives in scratch memory — a microsecond per field access) */
struct Tracer {
    const sre_scan_tables_t *T;         /* cls / trans / fast point into LDS */
    const sre_seg_summary_t *sum;       /* this stream's summaries */
    const uint8_t           *data;
    int64_t                  n, sp;
    uint32_t                 seg_bytes, init_state;
    int64_t                  apos;      /* anchor: state `astate` holds before position apos (-1 none) */
    uint32_t                 astate;
    uint16_t                *ck;        /* checkpoint states of the loaded segment, every 64 bytes */
    __attribute__((address_space(3))) uint16_t *trace;  /* states before each position of the loaded 64-byte block (LDS) */
    __attribute__((address_space(3))) uint8_t  *syms;   /* ... and the byte classes at them: a walk step reads no global memory */
    int64_t                  seg_lo, seg_hi;    /* loaded segment: [seg_lo, seg_hi], -1 none */
    int64_t                  blk_lo, blk_hi;
    int64_t                  valid_from;        /* see entry_state */
    bool                     use_stable;        /* the summaries' stable stretches may be used (FIRST) */
    uint32_t                 seg_entry;         /* state before seg_lo */
    int64_t                  su_k = -1, su_lo = 0, su_hi = 0;   /* segment [su_lo, su_hi) whose stable prefix is cached below */
    uint32_t                 su_until = 0, su_s_in = 0;

#ifdef SRE_DEBUG_WALK
    unsigned long long dbg_seg_ticks = 0, dbg_blk_ticks = 0, dbg_entry_ticks = 0, dbg_top = 0, dbg_sb = 0, dbg_rest = 0;
    uint32_t           dbg_seg_calls = 0, dbg_blk_calls = 0, dbg_steps = 0;
#endif
    /* the hot tables as LDS-typed pointers in registers (bind() after T is set; see LineageWalk::run) */
    const __attribute__((address_space(3))) sre_dev_trans_t *ltrans;
    const __attribute__((address_space(3))) uint8_t         *lcls;
    uint32_t                                                  nsym;
    __device__ inline void bind()
    {
        ltrans = (const __attribute__((address_space(3))) sre_dev_trans_t *) (uintptr_t) T->trans;
        lcls = (const __attribute__((address_space(3))) uint8_t *) (uintptr_t) T->cls;
        nsym = T->ncls + 1;
    }

    __device__ __forceinline__ uint32_t step(uint32_t st, int64_t q) const
    {
        return ltrans[(size_t) st * nsym + lcls[data[q]]].next;
    }

    /* the state chain over [lo, lo + cnt), cnt <= 64, in two passes: first the byte classes
     * (independent loads from global memory: they overlap), then the chain through LDS only
     * — one lane: a load inside the chain costs its whole latency per byte.  Uses `syms` (and,
     * with record, `trace`: the state before every position, and behind the last). */
    __device__ __forceinline__ uint32_t run_block(uint32_t cur, int64_t lo, uint32_t cnt, bool record)
    {
#ifdef SRE_DEBUG_WALK
        const unsigned long long dbg_t = wall_clock64();
        dbg_blk_calls++;
#endif
        /* GLOBAL loads (not flat ones: those also count on the LDS counter and would be
         * waited for one by one): a whole aligned block as four 16-byte loads in flight
         * together, else eight byte loads at a time.  Beside a resident scan a round trip to
         * HBM takes microseconds; the walk should make a handful, not one per byte. */
        const __attribute__((address_space(1))) uint8_t *g =
            (const __attribute__((address_space(1))) uint8_t *) (uintptr_t) (data + lo);
        if (cnt == 64 && (reinterpret_cast<uintptr_t>(data + lo) & 15) == 0) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const __attribute__((address_space(1))) u32x4 *gv = (const __attribute__((address_space(1))) u32x4 *) g;
            const u32x4 v0 = gv[0], v1 = gv[1], v2 = gv[2], v3 = gv[3];
            const uint32_t w[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w,
                                    v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (uint32_t x = 0; x < 64; x++) syms[x] = lcls[(w[x >> 2] >> ((x & 3) * 8)) & 0xffu];
        } else {
#pragma unroll 8
            for (uint32_t x = 0; x < cnt; x++) syms[x] = lcls[g[x]];
        }
        for (uint32_t x = 0; x < cnt; x++) {
            if (record) trace[x] = (uint16_t) cur;
            cur = ltrans[(size_t) cur * nsym + syms[x]].next;
        }
        if (record) trace[cnt] = (uint16_t) cur;
#ifdef SRE_DEBUG_WALK
        dbg_blk_ticks += wall_clock64() - dbg_t;
#endif
        return cur;
    }

    /* 16 bytes at a 16-byte aligned position through the packed-class fast
     * table; groups holding an event fall back to byte steps */
    __device__ __forceinline__ uint32_t step16(uint32_t st, int64_t q) const
    {
        const uint32_t bits = T->class_bits, stride = T->stride;
        const uint4    v = *reinterpret_cast<const uint4 *>(data + q);
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
        uint32_t       so = st * SRE_FAST_ROW_BYTES, acc = 0;
        for (uint32_t j = 0; j < 16 / stride; j++) {
            uint32_t idx = 0;
            for (uint32_t u = 0; u < stride; u++) {
                const uint32_t b = j * stride + u;
                const uint32_t c = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
                idx |= (bits == 8 ? c : (uint32_t) T->cls[c]) << (u * bits);
            }
            const uint32_t t = T->fast[(so >> 2) + idx];
            acc |= t;
            so = t & ~(SRE_FAST_ROW_BYTES - 1);
        }
        if (acc & SRE_FAST_SLOW) {
            for (int b = 0; b < 16; b++) st = step(st, q + b);
            return st;
        }
        return so / SRE_FAST_ROW_BYTES;
    }

    /*
     * State of THIS search (the one that started at sp) before the first byte
     * of segment kq > sp's segment.  The entry state a scan lane recorded
     * (s_in) is the state in scan order; it belongs to this search only from
     * the segment behind the one in which the search was started in scan order:
     * in COUNT mode the previous match may have been completed segments later
     * than it ended (its list lived on), and the lane that completed it then
     * walked back to sp — the entry states of the segments in between belong
     * to the older search.  `valid_from` (first_valid_segment above) is the
     * first segment whose s_in can be taken; in front of it the state is
     * replayed from sp.
     */
    __device__ __forceinline__ uint32_t entry_state(int64_t kq) const
    {
        if (kq >= valid_from) return sum[kq].s_in & ~SRE_STATE_SKIP;
        uint32_t cur = init_state;
        for (int64_t q = sp; q < kq * (int64_t) seg_bytes; q++) cur = step(cur, q);
        return cur;
    }

    /* checkpoint segment kq up to (and including the block of) position upto */
    __device__ __forceinline__ void load_segment(int64_t kq, int64_t upto)
    {
#ifdef SRE_DEBUG_WALK
        const unsigned long long dbg_t = wall_clock64();
        dbg_seg_calls++;
#endif
        int64_t  lo = kq * (int64_t) seg_bytes, hi = lo + seg_bytes;
        uint32_t cur;
        if (hi > n) hi = n;
        if (((upto + 64) & ~(int64_t) 63) < hi) hi = (upto + 64) & ~(int64_t) 63;
        if (sp >= lo) {
            lo = sp;
            cur = init_state;
        } else {
#ifdef SRE_DEBUG_WALK
            const unsigned long long dbg_te = wall_clock64();
#endif
            cur = entry_state(kq);
#ifdef SRE_DEBUG_WALK
            dbg_entry_ticks += wall_clock64() - dbg_te;
#endif
            if (use_stable && kq >= valid_from) {
                /* the state does not move over the stable prefix: replay from its end */
                int64_t skip = (int64_t) (sum[kq].stable_until & ~63u);
                if (skip > upto - lo) skip = (upto - lo) & ~(int64_t) 63;
                if (skip > 0 && lo + skip < hi) lo += skip;
            }
        }
        seg_lo = lo;
        seg_hi = hi;
        seg_entry = cur;
        /* ck[i] = state before position c0 + 64 * i, c0 = lo rounded up to 64 */
        const int64_t c0 = (lo + 63) & ~(int64_t) 63;
        int64_t       q = lo;
        const bool    aligned = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
        if (q < c0 && q < hi) {
            const int64_t stop = c0 < hi ? c0 : hi;
            cur = run_block(cur, q, (uint32_t) (stop - q), false);
            q = stop;
        }
        uint32_t i = 0;
        while (q < hi) {
            ck[i++] = (uint16_t) cur;
            const int64_t stop = q + 64 < hi ? q + 64 : hi;
            if (aligned && stop - q == 64 && hi - q > 256) {
                /* (long replays: the packed fast table, 16 bytes a load) */
                for (int g4 = 0; g4 < 4; g4++) cur = step16(cur, q + 16 * g4);
            } else {
                cur = run_block(cur, q, (uint32_t) (stop - q), false);
            }
            q = stop;
        }
        ck[i] = (uint16_t) cur;
        blk_lo = 1;
        blk_hi = 0;
#ifdef SRE_DEBUG_WALK
        dbg_seg_ticks += wall_clock64() - dbg_t;
#endif
    }

    /* state before position q (sp <= q <= n) */
    __device__ __forceinline__ uint32_t state_before(int64_t q)
    {
        if (q == sp) return init_state;
        if (use_stable && q > sp) {
            /* inside the stable prefix of its segment the state is the entry state */
            if (q - 1 < su_lo || q - 1 >= su_hi) {
                /* (a 64-bit division and two reads of global memory: once per segment, not per
                 * step — the walk is one lane, every instruction of it is latency) */
                su_k = (q - 1) / seg_bytes;
                su_lo = su_k * (int64_t) seg_bytes;
                su_hi = su_lo + seg_bytes;
                su_until = sum[su_k].stable_until;
                su_s_in = sum[su_k].s_in;
            }
            const int64_t kq = su_k, sbase = su_lo;
            if (sbase >= sp && kq >= valid_from && q - sbase <= (int64_t) su_until) {
                return su_s_in & ~SRE_STATE_SKIP;
            }
        }
        if ((q < blk_lo || q > blk_hi) && apos >= 0 && q > apos && q <= apos + 64) {
            /* the block right behind the anchor: no segment replay needed */
            const int64_t hi = apos + 64 < n ? apos + 64 : n;
            run_block(astate, apos, (uint32_t) (hi - apos), true);
            blk_lo = apos;
            blk_hi = hi;
        }
        if (q < blk_lo || q > blk_hi) {
            if (q <= seg_lo || q > seg_hi) load_segment((q - 1) / seg_bytes, q);
            /* 64-byte block holding q - 1 and q */
            const int64_t c0 = (seg_lo + 63) & ~(int64_t) 63;
            int64_t       lo = (q - 1) & ~(int64_t) 63, hi;
            uint32_t      cur;
            if (lo < c0) {
                /* in front of the first checkpoint: replay from the segment entry */
                lo = seg_lo;
                cur = seg_entry;
                hi = c0 < seg_hi ? c0 : seg_hi;
            } else {
                cur = ck[(lo - c0) / 64];
                hi = lo + 64 < seg_hi ? lo + 64 : seg_hi;
            }
            run_block(cur, lo, (uint32_t) (hi - lo), true);
            blk_lo = lo;
            blk_hi = hi;
        }
        return trace[q - blk_lo];
    }

    /* class of the byte at position q (the loaded block knows it) */
    __device__ __forceinline__ uint32_t sym_at(int64_t q) const
    {
        if (q >= blk_lo && q < blk_hi) return syms[q - blk_lo];
        return lcls[data[q]];
    }
};

/*
 * Ancestor maps (sre_dfa.h lineage, in parallel): one lane per segment in front
 * of a flagged stream's match walks its segment forwards from the verified
 * entry state and tracks, for every thread of the current list, the index of
 * its ancestor in the segment's entry list plus two sticky bits (a SAVE on the
 * way; passage through the ".*?" thread / a skip re-seed).  The capture walker
 * can then jump over whole segments — and, through the 256-segment
 * compositions, over whole blocks — in which its lineage did nothing.
 */
__global__ __launch_bounds__(256) void
sre_k_lineage_maps(const sre_scan_tables_t *__restrict__ tabp, sre_scan_geom_t G,
                   const sre_seg_summary_t *__restrict__ sum,
                   const sre_stream_status_t *__restrict__ status,
                   sre_seg_lineage_t *__restrict__ maps)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const sre_scan_tables_t &T = *tabp;
    const uint32_t nsym = T.ncls + 1;
    /* LDS copies: class map, transition records, parent + flag bytes */
    uint8_t         *clsl = lds;
    sre_dev_trans_t *trl = reinterpret_cast<sre_dev_trans_t *>(lds + 256);
    const uint32_t   ntr = T.nstates * nsym;
    uint8_t         *parl = reinterpret_cast<uint8_t *>(trl + ntr);
    uint8_t         *flgl = parl + ((T.lin_total + 15u) & ~15u);
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) clsl[i] = T.cls[i];
    for (uint32_t i = threadIdx.x; i < ntr; i += blockDim.x) trl[i] = T.trans[i];
    for (uint32_t i = threadIdx.x; i < T.lin_total; i += blockDim.x) {
        parl[i] = T.lin_parent[i];
        flgl[i] = T.lin_flags[i];
    }
    __syncthreads();

    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G.nsegs) return;
    const uint32_t s = stream_of(G, g);
    const sre_stream_status_t &st = status[s];
    if (!st.need_maps) return;
    const uint64_t first = geom_first(G, s);
    const int64_t  k = (int64_t) (g - first);
    const int64_t  sp = st.ev_sp;
    if (k > st.ev_seg || (k + 1) * (int64_t) G.seg_bytes <= sp) return;

    const uint8_t *data = geom_ptr(G, s);
    const int64_t  n = (int64_t) geom_len(G, s);
    int64_t        lo = k * (int64_t) G.seg_bytes, hi = lo + G.seg_bytes;
    uint32_t       cur = sum[g].s_in & ~SRE_STATE_SKIP;
    if (hi > n) hi = n;
    if (k == st.ev_seg && st.ev_pos < hi) hi = st.ev_pos;      /* list at the event position */
    if (sp >= lo) {
        lo = sp;
        cur = T.init[sp == 0 ? G.init_variant : restart_variant(T, data, sp)];
    } else if (k < (int64_t) st.valid_from) {
        /* the recorded entry state belongs to an older search (Tracer::entry_state) */
        cur = T.init[sp == 0 ? G.init_variant : restart_variant(T, data, sp)];
        for (int64_t x = sp; x < lo; x++) cur = trl[cur * nsym + clsl[data[x]]].next;
    }

    uint64_t anc = 0xfedcba9876543210ull;       /* identity */
    uint32_t saved = 0, stop = 0;
    uint32_t last_off = 0xffffffffu;            /* lineage vector applied by the previous byte */
    auto step = [&](uint32_t c) {
        const sre_dev_trans_t &tr = trl[cur * nsym + clsl[c]];
        if (tr.pad && tr.lin_off == last_off) {
            /* the same idempotent map again (a list looping in place): no change */
            cur = tr.next;
            return;
        }
        last_off = tr.lin_off;
        uint64_t nanc = 0;
        uint32_t nsaved = 0, nstop = 0;
        for (uint32_t j = 0; j < tr.lin_n; j++) {
            const uint32_t par = parl[tr.lin_off + j], f = flgl[tr.lin_off + j];
            if (par == 0xffu) {
                nstop |= 1u << j;
            } else {
                nanc |= ((anc >> (4 * par)) & 15ull) << (4 * j);
                nsaved |= (((saved >> par) & 1u) | (f & 1u)) << j;
                nstop |= (((stop >> par) & 1u) | ((f >> 1) & 1u)) << j;
            }
        }
        anc = nanc;
        saved = nsaved;
        stop = nstop;
        cur = tr.next;
    };
    int64_t q = lo;
    if ((reinterpret_cast<uintptr_t>(data) & 15) == 0) {
        for (; q < hi && (q & 63); q++) step(data[q]);
        /* 64 bytes per turn, the next 64 already in flight */
        uint4 nx[4];
        if (q + 64 <= hi) {
#pragma unroll
            for (int x = 0; x < 4; x++) nx[x] = *reinterpret_cast<const uint4 *>(data + q + 16 * x);
        }
        for (; q + 64 <= hi; q += 64) {
            uint4 v[4];
#pragma unroll
            for (int x = 0; x < 4; x++) v[x] = nx[x];
            if (q + 128 <= hi) {
#pragma unroll
                for (int x = 0; x < 4; x++) nx[x] = *reinterpret_cast<const uint4 *>(data + q + 64 + 16 * x);
            }
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint32_t words[4] = {v[x].x, v[x].y, v[x].z, v[x].w};
#pragma unroll
                for (int b = 0; b < 16; b++) step((words[b >> 2] >> ((b & 3) * 8)) & 0xffu);
            }
        }
    }
    for (; q < hi; q++) step(data[q]);
    sre_seg_lineage_t out;
    out.anc = anc;
    out.saved = (uint16_t) saved;
    out.stop = (uint16_t) stop;
    out.pad = 0;
    maps[g] = out;
}

/* compose SRE_LINEAGE_BLOCK consecutive segment maps (global segment ids):
 * 16 adjacent lanes per block of segments, one per thread index */
__global__ __launch_bounds__(256) void
sre_k_lineage_blocks(uint64_t nsegs, const sre_seg_lineage_t *__restrict__ maps,
                     sre_seg_lineage_t *__restrict__ blocks)
{
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t b = t / 16;
    const uint32_t j = (uint32_t) (t % 16);
    const uint64_t g0 = b * SRE_LINEAGE_BLOCK;
    const bool     live = g0 + SRE_LINEAGE_BLOCK <= nsegs;
    uint32_t       cur = j;
    bool           bad = false;
    if (live) {
        for (int64_t g = (int64_t) (g0 + SRE_LINEAGE_BLOCK) - 1; g >= (int64_t) g0 && !bad; g--) {
            const sre_seg_lineage_t &m = maps[g];
            if (((m.saved | m.stop) >> cur) & 1u) bad = true;
            else cur = (uint32_t) ((m.anc >> (4 * cur)) & 15ull);
        }
    }
    /* gather the 16 results of a block of segments in its first lane */
    uint64_t anc = (uint64_t) cur << (4 * j);
    uint32_t blocked = bad ? 1u << j : 0u;
    for (int d = 8; d >= 1; d >>= 1) {
        anc |= __shfl_down(anc, d, 16);
        blocked |= __shfl_down(blocked, d, 16);
    }
    if (live && j == 0) {
        sre_seg_lineage_t out;
        out.anc = anc;
        out.saved = (uint16_t) blocked;
        out.stop = 0;
        out.pad = 0;
        blocks[b] = out;
    }
}

/*
 * The backward lineage walk shared by the capture kernel (from the final match event)
 * and the streaming tail kernel (from every thread still listed at the end of a chunk):
 * thread j lives in the list at position p0; follow parent links backwards collecting
 * the capture slots still unresolved, crossing stable stretches and (with maps) whole
 * segments in O(1).  `carried`: the search began in FRONT of this buffer (a chunk of a
 * stream): a lineage that reaches offset 0 takes the rest of its slots from the vector
 * the context carries for that thread.  Values are buffer offsets + base.
 * Returns 1 when the plain walk ran out of budget and the ancestor maps are wanted.
 */
struct LineageWalk {
    const sre_scan_tables_t *T;         /* tables in LDS */
    const sre_scan_tables_t *tabp;      /* ... and in global memory (rarely used parts) */
    Tracer                  *tr;
    uint32_t                 variant;
    int64_t                  ev_seg, unst_seg;
    int64_t                  unst_end;      /* last segment that is not stable, -1 none; a huge value: unknown */
    uint64_t                 first;
    const sre_seg_lineage_t *maps, *blocks;
    bool                     use_maps;
    const int64_t           *carried;   /* [SRE_STREAM_MAX_THREADS][SRE_STREAM_MAX_SLOTS], or null */
    int64_t                  base;
    int64_t                  walk_budget;   /* positions of plain walk before the maps are asked for */

    __device__ __forceinline__ int run(int64_t p0, uint32_t j, uint32_t state_at_p0, uint64_t &unresolved, int64_t *vec)
    {
        const sre_scan_tables_t &T = *this->T;
        Tracer                  &tr = *this->tr;
        const uint32_t           nsym = T.ncls + 1;
        /* The walk is ONE lane: every load is pure latency.  The tables are in LDS
         * (stage_walk_tables) but reached through generic pointers kept in a struct that is in
         * LDS itself — two dependent flat loads per access, ~1.5 us per step; as LDS-typed
         * pointers in registers a step is a handful of ds_reads. */
        typedef const __attribute__((address_space(3))) uint32_t        *l_u32;
        typedef const __attribute__((address_space(3))) uint64_t        *l_u64;
        typedef const __attribute__((address_space(3))) uint8_t         *l_u8;
        typedef const __attribute__((address_space(3))) sre_dev_trans_t *l_tr;
        const l_u32    L_list_off = (l_u32) (uintptr_t) T.list_off, L_list_pcs = (l_u32) (uintptr_t) T.list_pcs;
        const l_u64    L_lin_saves = (l_u64) (uintptr_t) T.lin_saves;
        const l_u8     L_lin_parent = (l_u8) (uintptr_t) T.lin_parent;
        const l_tr     L_trans = (l_tr) (uintptr_t) T.trans;
        const uint32_t nslots = T.nslots, nstates = T.nstates;
        const uint64_t *const lin_early = tabp->lin_early;      /* global memory; rare */
        const int64_t  seg = (int64_t) tr.seg_bytes;
    const int64_t  k_sp = tr.sp / seg;
    const bool     can_jump = use_maps && T.max_threads <= 16;
    int64_t        budget = walk_budget;

    /* stable stretches (sre_hip_scan.h SRE_FAST_STABLE): one search from offset 0 */
    const uint16_t *const neutral = (T.mode == 1 && T.nshadow) ? tabp->neutral : nullptr;

    struct { uint32_t flags, s_in, s_out, stable_until, stable_from; } S = {0, 0, 0, 0, 0};
    int64_t S_k = -1, S_lo = -((int64_t) 1 << 40);     /* nothing cached */

    /* thread j lives in the list at position p */
    for (int64_t p = p0; unresolved; p--) {
#ifdef SRE_DEBUG_WALK
        const unsigned long long dbg_a = wall_clock64();
#endif
        if (neutral != nullptr) {
            /* cross, in O(1), every stretch in which the automaton sat in one state and
             * thread j descended from itself without saving: the stable prefix of the
             * segment in front of p, whole runs of stable segments, a stable suffix */
            for (;;) {
                if (p <= tr.sp) break;
                if (p - 1 < S_lo || p - 1 >= S_lo + seg) {
                    /* (a division and the summary in global memory: once per segment, not per step) */
                    S_k = (p - 1) / seg;
                    S_lo = S_k * seg;
                    const sre_seg_summary_t &G = tr.sum[S_k];
                    S.flags = G.flags;
                    S.s_in = G.s_in;
                    S.s_out = G.s_out;
                    S.stable_until = G.stable_until;
                    S.stable_from = G.stable_from;
                }
                const int64_t kq = S_k, sbase = S_lo;
                if (p == sbase + seg && kq > unst_end && (S.flags & SRE_SUM_STABLE)) {
                    /* ... up to the end of the buffer (walks that start there) */
                    if (!((neutral[S.s_in & ~SRE_STATE_SKIP] >> j) & 1u)) break;
                    p = (unst_end + 1) * seg;
                    if (p < tr.sp) p = tr.sp;
                    continue;
                }
                if (p == sbase + seg && kq < ev_seg && kq > unst_seg && (S.flags & SRE_SUM_STABLE)) {
                    /* a whole run of stable segments, all in one state */
                    if (!((neutral[S.s_in & ~SRE_STATE_SKIP] >> j) & 1u)) break;
                    p = (unst_seg + 1) * seg;
                    if (p < tr.sp) p = tr.sp;
                    continue;
                }
                if (p - sbase <= (int64_t) S.stable_until) {
                    if (!((neutral[S.s_in & ~SRE_STATE_SKIP] >> j) & 1u)) break;
                    p = sbase > tr.sp ? sbase : tr.sp;
                    continue;
                }
                if (p == sbase + seg && (int64_t) S.stable_from < seg) {
                    if (!((neutral[S.s_out & ~SRE_STATE_SKIP] >> j) & 1u)) break;
                    p = sbase + S.stable_from;
                    if (p < tr.sp) p = tr.sp;
                    continue;
                }
                break;
            }
        }
        if (can_jump && p > tr.sp && p % seg == 0 && p < p0) {
            /* at a segment start: jump over the segments (blocks) in front of it
             * in which this lineage neither saved nor restarted */
            int64_t k2 = p / seg - 1;
            while (k2 >= k_sp) {        /* the search's own (partial) segment has a map too */
                const uint64_t g2 = first + (uint64_t) k2;
                if (g2 % SRE_LINEAGE_BLOCK == SRE_LINEAGE_BLOCK - 1
                    && k2 - (int64_t) SRE_LINEAGE_BLOCK > k_sp)
                {
                    const sre_seg_lineage_t &bm = blocks[g2 / SRE_LINEAGE_BLOCK];
                    if (!((bm.saved >> j) & 1u)) {
                        j = (uint32_t) ((bm.anc >> (4 * j)) & 15ull);
                        k2 -= SRE_LINEAGE_BLOCK;
                        continue;
                    }
                }
                const sre_seg_lineage_t &m = maps[g2];
                if (((m.saved | m.stop) >> j) & 1u) break;
                j = (uint32_t) ((m.anc >> (4 * j)) & 15ull);
                k2--;
            }
            p = k2 < k_sp ? tr.sp : (k2 + 1) * seg;
        }
        if (!can_jump && --budget < 0 && T.max_threads <= 16) {
            return 1;                       /* come back with the ancestor maps */
        }
#ifdef SRE_DEBUG_WALK
        const unsigned long long dbg_b = wall_clock64();
        tr.dbg_top += dbg_b - dbg_a;
#endif
        const uint32_t s_here = (p == p0) ? state_at_p0 : tr.state_before(p);
#ifdef SRE_DEBUG_WALK
        const unsigned long long dbg_c = wall_clock64();
        tr.dbg_sb += dbg_c - dbg_b;
#endif
#ifdef SRE_DEBUG_WALK
        tr.dbg_steps++;
#endif
#ifdef SRE_DEBUG_WALK_STEPS
        printf("walk p %lld s_here %u j %u pc %u unresolved %llx\n", (long long) p, s_here, j,
               T.list_pcs[T.list_off[s_here] + j], (unsigned long long) unresolved);
#endif
        if (L_list_pcs[L_list_off[s_here] + j] == 1) break;      /* the ".*?" ANY thread */
        l_tr    t;
        int64_t val;
        if (p == tr.sp && carried != nullptr) {
            /* the search came in from the previous chunk: the rest is what the context
             * carries for this thread */
            for (uint32_t q = 0; q < nslots; q++) {
                if ((unresolved >> q) & 1) vec[q] = carried[(size_t) j * SRE_STREAM_MAX_SLOTS + q];
            }
            unresolved = 0;
            break;
        }
        if (p == tr.sp) {
            t = &L_trans[(size_t) nstates * nsym + variant];   /* initial closure */
            val = tr.sp;
        } else {
            const uint32_t s_prev = tr.state_before(p - 1);     /* (may load another block: first) */
            t = &L_trans[(size_t) s_prev * nsym + tr.sym_at(p - 1)];
            val = p;
        }
        const uint32_t lin = t->lin_off + j;
        const uint64_t m = L_lin_saves[lin] & unresolved;
        if (m) {
            for (uint32_t q = 0; q < nslots; q++) {
                if ((m >> q) & 1) vec[q] = val + base;
            }
            unresolved &= ~m;
        }
        if (lin_early != nullptr) {
            /* written by a look-ahead splice before the byte was consumed */
            const uint64_t m2 = lin_early[lin] & unresolved;
            for (uint32_t q = 0; q < nslots; q++) {
                if ((m2 >> q) & 1) vec[q] = val - 1 + base;
            }
            unresolved &= ~m2;
        }
        if (p == tr.sp) break;
        j = L_lin_parent[lin];
        if (j == 0xffu) break;              /* re-seeded by the leading-byte skip */
#ifdef SRE_DEBUG_WALK
        tr.dbg_rest += wall_clock64() - dbg_c;
#endif
    }

        return 0;
    }
};

/* the walkers' tables in LDS: [fast][class map][transition records][lineage saves]
 * [list offsets][list pcs][lineage parents]; *Ts becomes a view onto them */
__device__ inline void
stage_walk_tables(const sre_scan_tables_t *__restrict__ tabp, uint8_t *lds, sre_scan_tables_t *Tsp)
{
    sre_scan_tables_t &Ts = *Tsp;
    {
        /* the walker's tables live in LDS:
         * [fast][class map][transition records][lineage saves][lineage parents] */
        uint32_t *fast = reinterpret_cast<uint32_t *>(lds);
        uint8_t  *clsl = lds + tabp->fast_bytes;
        uint8_t  *trl = clsl + 256;
        const uint32_t tr_bytes = (tabp->nstates * (tabp->ncls + 1) + SRE_SCAN_NINIT) * (uint32_t) sizeof(sre_dev_trans_t);
        uint64_t *savl = reinterpret_cast<uint64_t *>(trl + tr_bytes);
        uint32_t *lofl = reinterpret_cast<uint32_t *>(savl + tabp->lin_total);
        uint32_t *lpcl = lofl + tabp->nstates + 1;
        uint8_t  *parl = reinterpret_cast<uint8_t *>(lpcl + tabp->list_total);
        if (threadIdx.x == 0) {
            Ts = *tabp;
            Ts.fast = fast;
            Ts.cls = clsl;
            Ts.trans = reinterpret_cast<const sre_dev_trans_t *>(trl);
            Ts.lin_saves = savl;
            Ts.lin_parent = parl;
            Ts.list_off = lofl;
            Ts.list_pcs = lpcl;
        }
        for (uint32_t i = threadIdx.x; i <= tabp->nstates; i += blockDim.x) lofl[i] = tabp->list_off[i];
        for (uint32_t i = threadIdx.x; i < tabp->list_total; i += blockDim.x) lpcl[i] = tabp->list_pcs[i];
        for (uint32_t i = threadIdx.x; i < tabp->lin_total; i += blockDim.x) {
            savl[i] = tabp->lin_saves[i];
            parl[i] = tabp->lin_parent[i];
        }
        for (uint32_t i = threadIdx.x; i < tabp->fast_bytes / 16; i += blockDim.x) {
            reinterpret_cast<uint4 *>(fast)[i] = reinterpret_cast<const uint4 *>(tabp->fast_plain)[i];
        }
        for (uint32_t i = threadIdx.x; i < tr_bytes / 8; i += blockDim.x) {
            reinterpret_cast<uint64_t *>(trl)[i] = reinterpret_cast<const uint64_t *>(tabp->trans)[i];
        }
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) clsl[i] = tabp->cls[i];
        __syncthreads();
    }
}

/* NT == 64: one wave per stream behind sre_launch_verify.  NT == 1024: ONE stream of at most
 * SRE_VERIFY_ONE_SEGS segments, FIRST / Thompson — the workgroup runs the chain check itself
 * (verify_one_stream), then lane 0 walks: a small buffer costs two launches, not six. */
template <int NT>
__global__ __launch_bounds__(NT) void
sre_k_captures(const sre_scan_tables_t *__restrict__ tabp, sre_scan_geom_t G,
               const sre_seg_summary_t *__restrict__ sum,
               sre_stream_status_t *__restrict__ status, uint16_t *__restrict__ scratch,
               int64_t *__restrict__ records, uint32_t ovec_slots,
               const sre_seg_lineage_t *__restrict__ maps,
               const sre_seg_lineage_t *__restrict__ blocks, int use_maps)
{
    SRE_TAIL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ sre_scan_tables_t Ts;
    __shared__ uint16_t sh_trace[64][72];   /* the walker's current 64-byte block (Tracer), per walking lane */
    __shared__ uint8_t  sh_syms[64][72];
#ifdef SRE_DEBUG_WALK
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    stage_walk_tables(tabp, lds, &Ts);
    if (NT != 64) {
        __shared__ VerifyAcc sh_acc;
        __shared__ sre_stream_status_t sh_st;
        verify_one_stream<NT>(Ts, sum, geom_first(G, 1), status, &sh_acc, &sh_st, G.digest);
    }
    /* one WORKGROUP per stream, lane 0 walks: the walks of a batch run side by side on
     * different CUs instead of as 64 divergent lanes of one wave (128 streams: 51 us -> see
     * profiles/r02_experiments.txt) */
    /* A batch of very many streams (a million log lines) is walked by LANES — 64 streams per wave, divergent but
     * side by side, on a bounded grid whose workgroups take their streams in turn: a walk is ~18 us of latency
     * on one lane, and 2048 resident workgroups of one walking lane each made 8.8 ms of 1M x 96 B with a match
     * in every line. */
    const bool     lane_mode = NT == 64 && G.nstreams >= SRE_CAPTURE_LANE_STREAMS;
    if (!lane_mode && threadIdx.x != 0) return;
    const uint32_t wl = lane_mode ? threadIdx.x : 0u;
    auto one_stream = [&](const uint32_t s) {
    if (use_maps && !status[s].need_maps) return;   /* second pass: flagged streams only */
#ifdef SRE_DEBUG_WALK
    const unsigned long long dbg_ts = wall_clock64();
    unsigned long long       dbg_tw0 = 0, dbg_tw1 = 0;
#endif
    const sre_scan_tables_t   &T = Ts;
    const sre_stream_status_t  st = status[s];
    int64_t                   *rec = records + (size_t) s * (2 + ovec_slots);
    const uint32_t             nsym = T.ncls + 1;

    status[s].need_maps = 0;
    rec[0] = st.rc;
    rec[1] = st.count;
    for (uint32_t q = 0; q < ovec_slots; q++) rec[2 + q] = -1;
    if (!st.done) {
        rec[0] = RC_ERROR;
        return;
    }
    if (st.error && T.mode == SRE_HIP_PIKE_COUNT) rec[0] = RC_ERROR;   /* iteration ended with SRE_ERROR */
    if (st.ev_pos < 0) return;
    if (T.mode == 0) {
        rec[0] = 0;                          /* Thompson: SRE_OK, no captures */
        return;
    }

    Tracer tr;
    tr.T = &T;
    tr.sum = sum + geom_first(G, s);
    tr.data = geom_ptr(G, s);
    tr.n = (int64_t) geom_len(G, s);
    tr.sp = st.ev_sp;
    tr.seg_bytes = G.seg_bytes;
    const uint32_t variant = st.ev_sp == 0 ? G.init_variant : restart_variant(T, tr.data, st.ev_sp);
    tr.init_state = T.init[variant];
    tr.apos = (st.ev_apos >= st.ev_sp) ? st.ev_apos : -1;
    tr.astate = st.ev_astate;
    tr.ck = scratch + (size_t) s * (G.seg_bytes + 16);
    tr.trace = (__attribute__((address_space(3))) uint16_t *) sh_trace[wl];
    tr.syms = (__attribute__((address_space(3))) uint8_t *) sh_syms[wl];
    tr.bind();
    tr.seg_lo = tr.seg_hi = -1;
    tr.blk_lo = 1;
    tr.blk_hi = 0;
    tr.seg_entry = 0;
    tr.valid_from = first_valid_segment(tr.sum, tr.sp, (int64_t) G.seg_bytes, st.ev_seg);
    tr.use_stable = (T.mode == 1 && T.nshadow != 0);
    status[s].valid_from = (uint32_t) tr.valid_from;

    int64_t  vec[64];
    uint64_t unresolved = T.nslots >= 64 ? ~0ull : ((1ull << T.nslots) - 1);
    for (uint32_t q = 0; q < T.nslots; q++) vec[q] = -1;

    const sre_dev_trans_t &te = T.trans[(size_t) st.ev_state * nsym + st.ev_sym];
    uint32_t               j = te.src;
    if (te.kind == EV_DONE || te.kind == SRE_DEV_EV_DONE_EMPTY) {
        const uint64_t m = te.saves & unresolved;
        for (uint32_t q = 0; q < T.nslots; q++) {
            if ((m >> q) & 1) vec[q] = st.ev_pos + 1;
        }
        unresolved &= ~m;
    }
    if (te.early & unresolved) {
        /* SAVEs of a look-ahead splice in front of the event: the position itself */
        const uint64_t m = te.early & unresolved;
        for (uint32_t q = 0; q < T.nslots; q++) {
            if ((m >> q) & 1) vec[q] = st.ev_pos;
        }
        unresolved &= ~m;
    }

    {
        LineageWalk lw;
        lw.T = &T;
        lw.tabp = tabp;
        lw.tr = &tr;
        lw.variant = variant;
        lw.ev_seg = st.ev_seg;
        lw.unst_seg = st.unst_seg;
        lw.unst_end = (int64_t) 1 << 62;        /* (the event's walk never starts behind its segment) */
        lw.first = geom_first(G, s);
        lw.maps = maps;
        lw.blocks = blocks;
        lw.use_maps = use_maps != 0;
        lw.carried = nullptr;
        lw.base = 0;
        lw.walk_budget = SRE_WALK_BUDGET;
#ifdef SRE_DEBUG_WALK
        dbg_tw0 = wall_clock64();
#endif
        const int need = lw.run(st.ev_pos, j, st.ev_state, unresolved, vec);
#ifdef SRE_DEBUG_WALK
        dbg_tw1 = wall_clock64();
#endif
        if (need) {
            status[s].need_maps = 1;         /* come back with the ancestor maps */
            return;
        }
    }

    /* slice the winning regex's groups (sre_vm_pike.c:945-989) */
    uint64_t ofs = 0;
    for (int64_t i = 0; i < st.rc; i++) ofs += T.multi_ncaps[i] + 1;
    ofs *= 2;
    const uint64_t ncopy = 2ull * (T.multi_ncaps[st.rc] + 1);
    for (uint64_t q = 0; q < ovec_slots; q++) rec[2 + q] = q < ncopy ? vec[ofs + q] : -1;
    rec[0] = (st.error && T.mode == SRE_HIP_PIKE_COUNT) ? RC_ERROR : st.rc;
#ifdef SRE_DEBUG_WALK
    printf("captures: stream %u (10 ns ticks) staging %llu, to the walk %llu, walk %llu, rest %llu; steps %u, load_segment %u calls %llu ticks "
           "(entry_state %llu), run_block %u calls %llu ticks; per step: top %llu, state_before(p) %llu, rest %llu\n", s,
           dbg_ts - dbg_t0, dbg_tw0 - dbg_ts, dbg_tw1 - dbg_tw0, wall_clock64() - dbg_tw1, tr.dbg_steps, tr.dbg_seg_calls,
           tr.dbg_seg_ticks, tr.dbg_entry_ticks, tr.dbg_blk_calls, tr.dbg_blk_ticks, tr.dbg_top, tr.dbg_sb, tr.dbg_rest);
#endif
    };
    if (lane_mode) {
        for (uint32_t s = blockIdx.x * 64u + wl; s < G.nstreams; s += gridDim.x * 64u) one_stream(s);
    } else {
        for (uint32_t s = blockIdx.x; s < G.nstreams; s += gridDim.x) one_stream(s);
    }
}


/*
 * Streaming (the chunked exec of the reference API, sre_vm_pike.c:148-689, on the
 * throughput path): ONE lane finishes one chunk of one stream after the scan and the
 * chain check.
 *   - a match event inside the chunk: its winner's capture vector is walked out now
 *     (lineage that reaches the chunk start continues in the vector the context
 *     carries for that thread) and becomes the pending match (:535-553);
 *   - the search is over (the list died, or eof): the pending match is the result
 *     (:607-636), or SRE_DECLINED at eof (:660-666);
 *   - otherwise SRE_AGAIN (:673-688): every thread still listed gets its capture
 *     vector resolved — that, with the automaton state, is what the next chunk needs —
 *     and the temporary match range is read off them (prepare_temp_captures,
 *     :692-735, offset quirk of :711/:721 kept).
 */
#define SRE_TAIL_THREADS 1024

/* verify != 0: the chain check of the chunk runs here too (verify_one_stream), in front of
 * the tail, and leaves the status word in status[0] */
__global__ __launch_bounds__(SRE_TAIL_THREADS) void
sre_k_stream_tail(const sre_scan_tables_t *__restrict__ tabp, sre_scan_geom_t G,
                  const sre_seg_summary_t *__restrict__ sum, sre_stream_status_t *__restrict__ status,
                  uint16_t *__restrict__ scratch, sre_stream_ctx_t *__restrict__ ctx,
                  sre_stream_result_t *__restrict__ res, int64_t base, int eof, uint32_t ovec_slots, int verify)
{
    SRE_TAIL_PRIO();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ sre_scan_tables_t Ts;
    __shared__ uint16_t sh_trace[72];       /* the walker's current 64-byte block (Tracer) */
    __shared__ uint8_t  sh_syms[72];
    __shared__ VerifyAcc sh_acc;
    __shared__ sre_stream_status_t sh_st;
#ifdef SRE_DEBUG_TAIL
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    stage_walk_tables(tabp, lds, &Ts);
#ifdef SRE_DEBUG_TAIL
    const unsigned long long dbg_t1 = wall_clock64();
#endif
    const uint64_t             nseg = geom_first(G, 1) - geom_first(G, 0);
    if (verify) verify_one_stream<SRE_TAIL_THREADS>(Ts, sum, nseg, status, &sh_acc, &sh_st, G.digest);
#ifdef SRE_DEBUG_TAIL
    const unsigned long long dbg_t2 = wall_clock64();
#endif
    if (threadIdx.x != 0) return;
    /* the host may be spinning on res->rc (host-visible memory): it is written last, behind
     * a system-scope fence, once everything else of the result is in place */
    int64_t rc_out = RC_ERROR;
    auto body = [&]() {
    const sre_scan_tables_t   &T = Ts;
    const sre_stream_status_t  st = verify ? sh_st : status[0];
    const uint32_t             nsym = T.ncls + 1;
    const bool                 continues = (G.flags & SRE_GEOM_CONTINUES) != 0;
    const int64_t              n = (int64_t) geom_len(G, 0);

    res->has_pending = 0;
    res->ev_in_chunk = 0;
    res->ev_slot1 = -1;
    res->poisoned = 0;
    res->next_state = 0;
    if (!continues) ctx->has_pending = 0;       /* a search starts with this chunk: nothing is carried in */
    if (!st.done) {
        rc_out = SRE_STREAM_UNSETTLED;
        return;
    }
    if (T.mode == 0) {
        /* Thompson (sre_vm_thompson.c:63-270): SRE_OK at the first MATCH thread met, else
         * SRE_DECLINED at eof / SRE_AGAIN with the list (= the state) carried on */
        /* A MATCH thread is met when the position it is listed at is RUN: one listed by the
         * closure behind the chunk's last byte waits for the next call that runs a position
         * (a byte, or the extra iteration at eof; :88, :265-269) */
        const bool carried_match = continues && ctx->has_pending != 0;
        bool       match_now = false, match_waits = false;
        if (carried_match) {
            match_now = n > 0 || eof;
            match_waits = !match_now;
        } else if (st.ev_pos >= 0) {
            const uint32_t kind = T.trans[(size_t) st.ev_state * (T.ncls + 1) + st.ev_sym].kind;
            const bool     listed_behind = (kind == EV_DONE || kind == SRE_DEV_EV_DONE_EMPTY);
            match_waits = listed_behind && st.ev_pos == n - 1 && !eof;
            match_now = !match_waits;
        }
        if (match_now) {
            rc_out = 0;
            ctx->state = 0;
            ctx->has_pending = 0;
        } else if (match_waits) {
            ctx->has_pending = 1;
            ctx->state = 1;                 /* (any live state: the next call does not get to use it) */
            res->next_state = 1;
            rc_out = -2;                    /* SRE_AGAIN */
        } else if (eof) {
            rc_out = RC_DECLINED;
            ctx->state = 0;
            ctx->has_pending = 0;
        } else {
            const uint32_t sF = sum[nseg - 1].s_out & ~SRE_STATE_SKIP;
            ctx->state = tabp->unskip[sF];
            ctx->has_pending = 0;
            res->next_state = ctx->state;
            rc_out = -2;                   /* SRE_AGAIN */
        }
        return;
    }

    Tracer tr;
    tr.T = &T;
    tr.sum = sum;
    tr.data = geom_ptr(G, 0);
    tr.n = n;
    tr.sp = 0;
    tr.seg_bytes = G.seg_bytes;
    const uint32_t variant = G.init_variant;
    tr.init_state = continues ? G.entry_state : T.init[variant];
    tr.apos = st.ev_apos >= 0 ? st.ev_apos : -1;
    tr.astate = st.ev_astate;
    tr.ck = scratch;
    tr.trace = (__attribute__((address_space(3))) uint16_t *) sh_trace;
    tr.syms = (__attribute__((address_space(3))) uint8_t *) sh_syms;
    tr.bind();
    tr.seg_lo = tr.seg_hi = -1;
    tr.blk_lo = 1;
    tr.blk_hi = 0;
    tr.seg_entry = 0;
    tr.valid_from = 1;                  /* one search per chunk: every recorded entry state is its own */
    tr.use_stable = (T.nshadow != 0);

    LineageWalk lw;
    lw.T = &T;
    lw.tabp = tabp;
    lw.tr = &tr;
    lw.variant = variant;
    lw.ev_seg = st.ev_seg >= 0 ? st.ev_seg : (int64_t) nseg;
    lw.unst_seg = st.ev_seg >= 0 ? st.unst_seg : (int64_t) nseg;
    lw.unst_end = T.nshadow ? st.unst_end : (int64_t) 1 << 62;
    lw.first = 0;
    lw.maps = lw.blocks = nullptr;
    lw.use_maps = false;
    lw.carried = continues ? &ctx->caps[0][0] : nullptr;
    lw.base = base;
    lw.walk_budget = (int64_t) 1 << 62;

    const uint64_t all = T.nslots >= 64 ? ~0ull : ((1ull << T.nslots) - 1);
    int64_t        vec[SRE_STREAM_MAX_SLOTS];

    /* ---- a match event inside this chunk replaces the pending match */
    if (st.ev_pos >= 0) {
        uint64_t unresolved = all;
        for (uint32_t q = 0; q < T.nslots; q++) vec[q] = -1;
        const sre_dev_trans_t &te = T.trans[(size_t) st.ev_state * nsym + st.ev_sym];
        if (te.kind == EV_DONE || te.kind == SRE_DEV_EV_DONE_EMPTY) {
            const uint64_t m = te.saves & unresolved;
            for (uint32_t q = 0; q < T.nslots; q++) {
                if ((m >> q) & 1) vec[q] = st.ev_pos + 1 + base;
            }
            unresolved &= ~m;
        }
        if (te.early & unresolved) {
            const uint64_t m = te.early & unresolved;
            for (uint32_t q = 0; q < T.nslots; q++) {
                if ((m >> q) & 1) vec[q] = st.ev_pos + base;
            }
            unresolved &= ~m;
        }
        /* carried stride: the context's rows are SRE_STREAM_MAX_SLOTS wide */
        lw.run(st.ev_pos, te.src, st.ev_state, unresolved, vec);
        ctx->has_pending = 1;
        ctx->pending_regex = st.rc;
        for (uint32_t q = 0; q < T.nslots; q++) ctx->pending_vec[q] = vec[q];
        res->ev_in_chunk = 1;
        res->ev_slot1 = vec[1];
    }

    /* ---- is the search over? */
    const uint64_t last = (uint64_t) st.limit >= 1 ? (uint64_t) st.limit - 1 : 0;
    const bool     died = (sum[last].flags & SRE_SUM_TERM) != 0;
    if (died || eof) {
        if (ctx->has_pending) {
            const int64_t id = ctx->pending_regex;
            uint64_t      ofs = 0;
            for (int64_t i = 0; i < id; i++) ofs += T.multi_ncaps[i] + 1;
            ofs *= 2;
            const uint64_t ncopy = 2ull * (T.multi_ncaps[id] + 1);
            for (uint64_t q = 0; q < ovec_slots && q < SRE_STREAM_MAX_SLOTS; q++) {
                res->ov[q] = q < ncopy ? ctx->pending_vec[ofs + q] : -1;
            }
            rc_out = id;
            res->poisoned = st.error ? 1 : 0;
        } else {
            rc_out = RC_DECLINED;
        }
        ctx->state = 0;
        ctx->has_pending = 0;
        return;
    }

    /* ---- SRE_AGAIN: carry the list (state + every thread's captures) to the next chunk */
    const uint32_t sF = sum[nseg - 1].s_out & ~SRE_STATE_SKIP;
    const uint32_t nth = T.list_off[sF + 1] - T.list_off[sF];
    /* what the next chunk starts from: a leading-byte skip that is still travelling ends
     * with the chunk (sre_dfa.h `unskip`); the list, and so the vectors, are the same */
    const uint32_t sNext = tabp->unskip[sF];
    int64_t        a0 = -1, a1 = -1;
    /* the new vectors go to a second set of rows first: the walk of thread j may still
     * read the carried vector of any thread */
    int64_t *fresh = &ctx->caps_next[0][0];
    for (uint32_t j = 0; j < nth && j < SRE_STREAM_MAX_THREADS; j++) {
        uint64_t unresolved = all;
        for (uint32_t q = 0; q < T.nslots; q++) vec[q] = -1;
        lw.run(n, j, sF, unresolved, vec);
        for (uint32_t q = 0; q < T.nslots; q++) fresh[(size_t) j * SRE_STREAM_MAX_SLOTS + q] = vec[q];
        /* prepare_temp_captures (:692-735) */
        uint64_t ofs = 0;
        for (uint32_t r = 0; r < T.nregexes; r++) {
            int64_t b = vec[ofs];
            if (b != -1 && (a0 == -1 || b < a0)) a0 = b;
            b = vec[1];
            if (b != -1 && (a1 == -1 || b > a1)) a1 = b;
            ofs += 2ull * (T.multi_ncaps[r] + 1);
        }
    }
    for (uint32_t j = 0; j < nth && j < SRE_STREAM_MAX_THREADS; j++) {
        for (uint32_t q = 0; q < T.nslots; q++) ctx->caps[j][q] = fresh[(size_t) j * SRE_STREAM_MAX_SLOTS + q];
    }
    ctx->state = sNext;
    res->next_state = sNext;
    res->ov[0] = a0;
    res->ov[1] = a1;
    rc_out = -2;                       /* SRE_AGAIN */
    if (ctx->has_pending) {
        const int64_t id = ctx->pending_regex;
        uint64_t      ofs = 0;
        for (int64_t i = 0; i < id; i++) ofs += T.multi_ncaps[i] + 1;
        ofs *= 2;
        res->has_pending = 1;
        res->pending[0] = ctx->pending_vec[ofs];
        res->pending[1] = ctx->pending_vec[ofs + 1];
    }
    };
    body();
#ifdef SRE_DEBUG_TAIL
    /* (10 ns ticks) table staging, chain check, the lane's work */
    res->ov[SRE_STREAM_MAX_SLOTS - 3] = (int64_t) (dbg_t1 - dbg_t0);
    res->ov[SRE_STREAM_MAX_SLOTS - 2] = (int64_t) (dbg_t2 - dbg_t1);
    res->ov[SRE_STREAM_MAX_SLOTS - 1] = (int64_t) (wall_clock64() - dbg_t2);
#endif
    __threadfence_system();
    *reinterpret_cast<volatile int64_t *>(&res->rc) = rc_out;
}

}  // namespace

typedef void (*sre_scan_kernel_t)(const sre_scan_tables_t *, sre_scan_geom_t, sre_seg_summary_t *,
                                  const sre_stream_status_t *, const uint8_t *);

template <int MODE, bool GROW>
static sre_scan_kernel_t
scan_kernel_bits(uint32_t bits, bool wide4)
{
    switch (bits) {
    case 1: return sre_k_scan<MODE, 1, true, GROW>;
    case 2: return sre_k_scan<MODE, 2, true, GROW>;
    case 4: return (MODE == SRE_HIP_PIKE_COUNT && wide4) ? sre_k_scan<MODE, 4, (MODE == SRE_HIP_PIKE_COUNT), GROW>
                                                         : sre_k_scan<MODE, 4, false, GROW>;
    default: return sre_k_scan<MODE, 8, false, GROW>;
    }
}

/* the variant that runs for these tables: [mode][class bits][tile index width][growing matches] */
static sre_scan_kernel_t
scan_kernel(const sre_scan_tables_t *h_tab)
{
    if (h_tab->mode != SRE_HIP_PIKE_COUNT) return scan_kernel_bits<1, false>(h_tab->class_bits, false);
    return h_tab->any_fresh ? scan_kernel_bits<SRE_HIP_PIKE_COUNT, true>(h_tab->class_bits, h_tab->wide != 0)
                            : scan_kernel_bits<SRE_HIP_PIKE_COUNT, false>(h_tab->class_bits, h_tab->wide != 0);
}

extern "C" size_t
sre_scan_lds_bytes(const sre_scan_tables_t *h_tab)
{
    const size_t tr = ((size_t) h_tab->nstates * (h_tab->ncls + 1) * 2 + 15) & ~(size_t) 15;
    /* index tile row: the two halves of a line as raw bytes (8 class bits) or
     * 16-bit scaled indices, plus the pad */
    const size_t half = (size_t) SRE_SCAN_ROUND * h_tab->class_bits / 8 * (h_tab->wide ? 2 : 1);
    /* [fast rows][class map][transitions][state flags][tile] */
    /* (COUNT runs at the three workgroups per CU its registers allow: an earlier "two are
     * faster" was an artefact of segment sizes that left a second, mostly empty round of
     * workgroups — profiles/r02_experiments.txt) */
    static const char *pad_env = getenv("SRE_HIP_LDS_PAD");      /* experiment knob: fewer workgroups per CU */
    const size_t pad = pad_env ? (size_t) atoi(pad_env) : 0;
    size_t       need = pad + (size_t) h_tab->fast_rows * SRE_FAST_ROW_BYTES + 256 + tr + ((h_tab->nstates + 15u) & ~15u)
                        + 16 + (size_t) SRE_SCAN_BLOCK * (2 * half + 16);
    return need + (size_t) SRE_SCAN_BLOCK * 16;      /* row descriptors */
}

/* workgroups of the scan kernel one CU can hold (registers and LDS), for the
 * geometry heuristic */
extern "C" int
sre_scan_blocks_per_cu(const sre_scan_tables_t *h_tab)
{
    int          n = 0;
    const size_t shmem = sre_scan_lds_bytes(h_tab);
    hipError_t   e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_kernel(h_tab), SRE_SCAN_BLOCK, shmem);
    if (e != hipSuccess || n < 1) n = 1;
    if (n > 8) n = 8;
    return n;
}

extern "C" hipError_t
sre_launch_scan(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab, sre_scan_geom_t geom,
                sre_seg_summary_t *d_sum, const sre_stream_status_t *d_lo, const uint8_t *d_entry,
                hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    const uint32_t grid = (uint32_t) ((geom.nsegs + SRE_SCAN_BLOCK - 1) / SRE_SCAN_BLOCK);
    const size_t   shmem = sre_scan_lds_bytes(&h_tab);
    if (shmem > 48 * 1024) {
        /* more than the default dynamic LDS limit: opt in (once per variant) */
        static bool raised[2][9][2];
        bool       &done = raised[h_tab.mode == SRE_HIP_PIKE_COUNT][h_tab.class_bits & 8 ? 8 : h_tab.class_bits][h_tab.wide != 0];
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan_kernel(&h_tab)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               SRE_SCAN_LDS_LIMIT);
            if (e != hipSuccess) return e;
            done = true;
        }
    }
    hipLaunchKernelGGL(scan_kernel(&h_tab), dim3(grid), dim3(SRE_SCAN_BLOCK), shmem, stream,
                       d_tab, geom, d_sum, d_lo, d_entry);
    return hipGetLastError();
}

/* the stream descriptors of a batch (pointers, lengths, first segments: 24 bytes a stream)
 * from pinned host memory to the device by a KERNEL: a DMA-engine copy in front of every
 * scan has to be synchronised with the compute queue, and on two alternating HIP streams
 * the first steps of a run stalled for milliseconds behind it */
__global__ __launch_bounds__(256) void
sre_k_upload_words(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}

extern "C" hipError_t
sre_launch_upload_words(const uint64_t *h_src_mapped, uint64_t *d_dst, uint32_t nwords, hipStream_t stream)
{
    if (nwords == 0) return hipSuccess;
    const uint32_t grid = (nwords + 255) / 256 < 64 ? (nwords + 255) / 256 : 64;
    hipLaunchKernelGGL(sre_k_upload_words, dim3(grid), dim3(256), 0, stream, h_src_mapped, d_dst, nwords);
    return hipGetLastError();
}

extern "C" size_t
sre_scan_verify_acc_bytes(uint32_t nstreams)
{
    return (size_t) nstreams * sizeof(VerifyAcc);
}

/* reset state of the accumulators: bad/end at ~0, the counters at 0 */
extern "C" hipError_t
sre_scan_verify_acc_init(void *d_acc, uint32_t nstreams, hipStream_t stream)
{
    hipError_t e = hipMemset2DAsync(d_acc, sizeof(VerifyAcc), 0xff, 2 * sizeof(unsigned long long),
                                    nstreams, stream);
    if (e != hipSuccess) return e;
    return hipMemset2DAsync(static_cast<char *>(d_acc) + 2 * sizeof(unsigned long long),
                            sizeof(VerifyAcc), 0, 5 * sizeof(unsigned long long), nstreams, stream);
}

extern "C" hipError_t
sre_launch_verify(sre_scan_tables_t h_tab, sre_scan_geom_t geom, const sre_seg_summary_t *d_sum,
                  void *d_acc, sre_stream_status_t *d_status, hipStream_t stream)
{
    if (geom.nstreams == 0) return hipSuccess;
    /* the accumulators are in their reset state: sre_scan_verify_acc_init at
     * allocation, sre_k_verify_c after every pass */
    VerifyAcc *acc = static_cast<VerifyAcc *>(d_acc);
    const uint32_t gseg = (uint32_t) ((geom.nsegs + 255) / 256);
    hipLaunchKernelGGL(sre_k_verify_a, dim3(gseg), dim3(256), 0, stream, geom, d_sum, acc, (int) h_tab.mode);
    const uint32_t gseg4 = (uint32_t) ((geom.nsegs + SRE_VERIFY_SPAN - 1) / SRE_VERIFY_SPAN);
    hipLaunchKernelGGL(sre_k_verify_b, dim3(gseg4), dim3(256), 0, stream, geom, d_sum, acc);
    if (h_tab.mode == 1 /* SRE_HIP_PIKE_FIRST */ && h_tab.nshadow) {
        hipLaunchKernelGGL(sre_k_verify_b2, dim3(gseg4), dim3(256), 0, stream, geom, d_sum, acc);
    }
    hipLaunchKernelGGL(sre_k_verify_c, dim3((geom.nstreams + 63) / 64), dim3(64), 0, stream, h_tab,
                       geom, d_sum, acc, d_status);
    return hipGetLastError();
}

extern "C" hipError_t
sre_launch_captures(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab, sre_scan_geom_t geom,
                    const sre_seg_summary_t *d_sum, sre_stream_status_t *d_status,
                    uint16_t *d_scratch, int64_t *d_records, uint32_t ovec_slots,
                    const sre_seg_lineage_t *d_maps, const sre_seg_lineage_t *d_blocks,
                    int use_maps, int verify, hipStream_t stream)
{
    if (geom.nstreams == 0) return hipSuccess;
    if (verify && (geom.nstreams != 1 || geom.nsegs > SRE_VERIFY_ONE_SEGS || h_tab.mode == SRE_HIP_PIKE_COUNT || use_maps)) {
        return hipErrorInvalidValue;
    }
    const uint32_t block = verify ? 1024 : 64;
    const uint32_t want = (!verify && geom.nstreams >= SRE_CAPTURE_LANE_STREAMS) ? (geom.nstreams + 63u) / 64u : geom.nstreams;
    const uint32_t grid = want < SRE_CAPTURE_MAX_GRID ? want : SRE_CAPTURE_MAX_GRID;
    const size_t   shmem = (size_t) h_tab.fast_bytes + 256
                         + ((size_t) h_tab.nstates * (h_tab.ncls + 1) + SRE_SCAN_NINIT) * sizeof(sre_dev_trans_t)
                         + (size_t) h_tab.lin_total * 9 + ((size_t) h_tab.nstates + 1 + h_tab.list_total) * 4 + 16;
    if (shmem > 48 * 1024) {
        /* a big automaton (many byte classes x states): one lane per stream, so the
         * workgroup may as well own most of the CU's LDS */
        static bool raised[2];
        if (!raised[verify != 0]) {
            hipError_t e = hipFuncSetAttribute(verify ? reinterpret_cast<const void *>(sre_k_captures<1024>)
                                                      : reinterpret_cast<const void *>(sre_k_captures<64>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, SRE_CAPTURE_LDS_LIMIT);
            if (e != hipSuccess) return e;
            raised[verify != 0] = true;
        }
    }
    if (verify) {
        hipLaunchKernelGGL(sre_k_captures<1024>, dim3(1), dim3(1024), shmem, stream, d_tab, geom, d_sum,
                           d_status, d_scratch, d_records, ovec_slots, d_maps, d_blocks, use_maps);
    } else {
        hipLaunchKernelGGL(sre_k_captures<64>, dim3(grid), dim3(block), shmem, stream, d_tab, geom, d_sum,
                           d_status, d_scratch, d_records, ovec_slots, d_maps, d_blocks, use_maps);
    }
    return hipGetLastError();
}

extern "C" hipError_t
sre_launch_lineage(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab, sre_scan_geom_t geom,
                   const sre_seg_summary_t *d_sum, const sre_stream_status_t *d_status,
                   sre_seg_lineage_t *d_maps, sre_seg_lineage_t *d_blocks, hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    const uint32_t nsym = h_tab.ncls + 1;
    const size_t   shmem = 256 + (size_t) h_tab.nstates * nsym * sizeof(sre_dev_trans_t)
                         + 2 * (size_t) ((h_tab.lin_total + 15u) & ~15u);
    const uint32_t gseg = (uint32_t) ((geom.nsegs + 255) / 256);
    if (shmem > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sre_k_lineage_maps),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, SRE_CAPTURE_LDS_LIMIT);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(sre_k_lineage_maps, dim3(gseg), dim3(256), shmem, stream, d_tab, geom, d_sum,
                       d_status, d_maps);
    const uint64_t nblocks = geom.nsegs / SRE_LINEAGE_BLOCK;
    if (nblocks) {
        hipLaunchKernelGGL(sre_k_lineage_blocks, dim3((uint32_t) ((nblocks * 16 + 255) / 256)), dim3(256), 0,
                           stream, geom.nsegs, d_maps, d_blocks);
    }
    return hipGetLastError();
}

/* FIRST / Thompson: exact entry states of every segment behind each unsettled stream's
 * verified prefix (see sre_k_seg_functions).  d_fn: nsegs * 64 bytes, d_comp:
 * (nsegs / 256 + 1) * 64, d_chunk_entry: nsegs / 256 + 1, d_entry: nsegs (out; 0xff =
 * not computed). */
extern "C" hipError_t
sre_launch_exact_entries(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab, sre_scan_geom_t geom,
                         const sre_seg_summary_t *d_sum, const sre_stream_status_t *d_status,
                         uint8_t *d_fn, uint8_t *d_comp, uint8_t *d_chunk_entry, uint8_t *d_entry,
                         hipStream_t stream)
{
    if (geom.nsegs == 0) return hipSuccess;
    const uint64_t nchunks = (geom.nsegs + SRE_FN_CHUNK - 1) / SRE_FN_CHUNK;
    hipError_t     e;
    if ((e = hipMemsetAsync(d_entry, 0xff, geom.nsegs, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_chunk_entry, 0xff, nchunks, stream)) != hipSuccess) return e;
    const size_t shmem = (size_t) (h_tab.nstates + 1) * SRE_FAST_ROW_BYTES + 256
                         + (((size_t) h_tab.nstates * (h_tab.ncls + 1) * 2 + 15) & ~(size_t) 15);
    const uint32_t grid = (uint32_t) ((geom.nsegs + 3) / 4);
    void (*kern)(const sre_scan_tables_t *, sre_scan_geom_t, const sre_stream_status_t *, uint8_t *) =
        h_tab.class_bits == 1 ? sre_k_seg_functions<1> : h_tab.class_bits == 2 ? sre_k_seg_functions<2>
        : h_tab.class_bits == 4 ? sre_k_seg_functions<4> : sre_k_seg_functions<8>;
    if (shmem > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int) shmem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, stream, d_tab, geom, d_status, d_fn);
    hipLaunchKernelGGL(sre_k_fn_chunks, dim3((uint32_t) nchunks), dim3(64), 0, stream, geom.nsegs, d_fn, d_comp);
    hipLaunchKernelGGL(sre_k_fn_resolve, dim3((geom.nstreams + 63) / 64), dim3(64), 0, stream, geom, d_sum,
                       d_status, d_fn, d_comp, d_entry, d_chunk_entry);
    hipLaunchKernelGGL(sre_k_fn_fill, dim3((uint32_t) ((nchunks + 63) / 64)), dim3(64), 0, stream, geom.nsegs,
                       d_fn, d_chunk_entry, d_entry);
    return hipGetLastError();
}

extern "C" hipError_t
sre_launch_stream_tail(const sre_scan_tables_t *d_tab, sre_scan_tables_t h_tab, sre_scan_geom_t geom,
                       const sre_seg_summary_t *d_sum, sre_stream_status_t *d_status,
                       uint16_t *d_scratch, sre_stream_ctx_t *d_ctx, sre_stream_result_t *result, int64_t base,
                       int eof, uint32_t ovec_slots, int verify, hipStream_t stream)
{
    const size_t shmem = (size_t) h_tab.fast_bytes + 256
                         + ((size_t) h_tab.nstates * (h_tab.ncls + 1) + SRE_SCAN_NINIT) * sizeof(sre_dev_trans_t)
                         + (size_t) h_tab.lin_total * 9 + ((size_t) h_tab.nstates + 1 + h_tab.list_total) * 4 + 16;
    if (shmem > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sre_k_stream_tail),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, SRE_CAPTURE_LDS_LIMIT);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(sre_k_stream_tail, dim3(1), dim3(SRE_TAIL_THREADS), shmem, stream, d_tab, geom, d_sum, d_status,
                       d_scratch, d_ctx, result, base, eof, ovec_slots, verify);
    return hipGetLastError();
}
