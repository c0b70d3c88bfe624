/*
 * sre_hip_pwave.hip — the exact Pike step taken by a WAVEFRONT (gfx950): lanes = the threads of
 * the ordered list, for the exact window behind the bit-parallel NFA scanner (sre_hip_nfa.hip).
 *
 * The one-lane VM (sre_hip_vm.hip) walks the reference loop thread by thread, ~1 us per
 * thread-step: the 13-57-byte windows of 19-60-thread programs cost 0.6-1.4 ms, more than the 4 GiB
 * set pass in front of them (profiles/r03_kernel_stats_nfa*.csv).  Here one byte step
 * (sre_vm_pike.c:312-581) is
 *   1. every lane i < n tests ITS thread against the byte (one LDS word of the thread's 256-bit
 *      accept map) and says whether it is a MATCH thread: two ballots — the consuming threads and
 *      the first listed MATCH, which cuts everything behind it (:530-553);
 *   2. the consuming threads in priority order (a scalar loop over the ballot's bits): the lanes
 *      load the source's STATIC closure list (sre_pwave.h), drop the targets an earlier source of
 *      this step has listed (a stamp per thread in LDS: the generation tags of :770, :792), rank
 *      the rest by a prefix count over the ballot of survivors — the new list's order is the
 *      source order, then the closure's order — and take the source's capture vector, the slots
 *      saved on the way replaced by the position (:826-837), broadcast from the source's column;
 *   3. a closure that reaches MATCH ends the step (SRE_DONE, :895-898): its captures become the
 *      match, lower-priority threads are dropped.
 * Between steps the leading-byte skip (:256-309) compares the list with the snapshot by one
 * ballot and looks for the next byte that can start a match 64 bytes at a time.
 * STABLE RUNS.  Captures never steer the VM, so a byte step is a function of (ordered list, seen_start_state,
 * byte).  A step that hands every thread its own place back — parent == itself, nothing saved on the
 * way, no MATCH — with seen_start_state as it was has changed NOTHING the later steps or the result can
 * see; its byte joins a 256-bit set that is valid while the list stands, and the bytes of that set that
 * follow are skipped 512 at a time (the reference's leading-byte skip, :256-309, for ANY list that loops
 * in place: [a-z]+ inside a word, .* behind its x, the idle .*? list of a program with too many leading
 * bytes for :992-1061).  Any other step drops the set.  tests/pwave_sim.cpp models it (on and off).
 * Everything the reference's quirks need is kept: seen_start_state, the snapshot without its last
 * thread, a skip target stepped without a check, the poisoned context.  Control flow is uniform;
 * the list (thread id per lane) and the capture vectors (one column per lane, slot-major) live in
 * LDS.  Cost per byte: ~2 LDS round trips per consuming thread plus 5 instructions per capture
 * slot and new thread — 0.3-0.5 us for the usual two or three consumers, a few us for thirty.
 * Measured (same box, bench.py --config nfa60, 4 GiB): whole step 1.33 ms with the one-lane
 * window, 0.98 ms with this kernel (the set pass alone: 0.94).
 */
#include <hip/hip_runtime.h>
#include "sre_pwave.h"
#include "sre_hip_common.h"
#include "sre_hip_vm.h"

#define PW_RC_DECLINED (-5)

namespace {

__device__ inline uint32_t
pw_lane_rank(uint64_t mask)
{
    /* set bits of `mask` below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

__device__ inline uint32_t
pw_uniform(uint32_t v)
{
    return (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
}

__device__ inline uint32_t
pw_readlane(uint32_t v, uint32_t lane)
{
    return (uint32_t) __builtin_amdgcn_readlane((int) v, (int) lane);
}

template <bool ENTS_LDS>
struct PikeWave {
    /* tables (LDS copies, entries possibly global) */
    const sre_pwave_hdr_t   *W;
    const uint32_t          *accw;      /* [64][8] */
    const sre_pwave_list_t  *lists;     /* [nlists][3] */
    const sre_pwave_entry_t *ents;      /* the entries in global memory */
    uint32_t                 ents_lds;  /* ... and their LDS copy's address when there is one (else ~0): a generic
                                         * pointer makes every entry load a flat_load */
    const uint16_t          *tid_list, *tid_match;
    /* state (LDS) */
    int64_t  *capsb;        /* two lists x [nslots][64]: list l at capsb + l * nslots * 64 (no member ARRAYS indexed at run time:
                             * they would send the whole struct to scratch memory — 240 bytes per lane and a memory round trip
                             * per member access, which is what the first build did: 1.5-2 us per byte) */
    int64_t  *matched;      /* [nslots] */
    uint16_t *tidvb;        /* two lists x [64] */
    uint32_t *stamp;        /* [64] */
    uint16_t *initial;      /* [64] */
    uint8_t  *inl;          /* [16] */
    uint32_t *stab;         /* [8] the stable-run byte set of the current list */
    uint32_t *lab;          /* two lists x [64]: equal labels => equal capture columns (copies of one column) */
    /* uniform state */
    const uint8_t *in;
    uint32_t nleading, nregexes;    /* (of the header: read once — a load per byte from global memory was the step's largest cost) */
    const uint32_t *lead;           /* [8] the leading-byte map, LDS copy */
    uint32_t win;           /* this lane's byte of the 64-byte input window [win_base, win_base + 64) */
    int64_t  win_base;
    uint32_t nslots, lane;
    uint32_t n, cur, stamp_cur;
    int64_t  processed;
    uint32_t seen_newline, sss, initial_count, has_matched, poisoned;
    uint32_t empty_capture, ctx_eof;    /* the re-armed context of a find-all iteration (:179-196, :616-628) */
    uint32_t first_buf, seen_word;      /* a context fed in chunks (exec_chunk) */
    uint32_t stab_on, stab_x, step_stable;
    int64_t  matched_id;

    __device__ __forceinline__ sre_pwave_entry_t entry(uint32_t idx) const
    {
        if (ENTS_LDS) {
            typedef const __attribute__((address_space(3))) uint16_t *lds_u16_t;
            typedef const __attribute__((address_space(3))) uint64_t *lds_u64_t;
            const uint32_t    a = ents_lds + idx * (uint32_t) sizeof(sre_pwave_entry_t);
            sre_pwave_entry_t e;
            e.tid = *reinterpret_cast<lds_u16_t>((uintptr_t) a);
            e.saves = *reinterpret_cast<lds_u64_t>((uintptr_t) (a + 8u));
            return e;
        }
        return ents[idx];
    }

    __device__ __forceinline__ size_t nslots_x64() const { return (size_t) nslots * 64u; }

    __device__ __forceinline__ uint32_t ctx_at(int64_t pos) const
    {
        if (pos == 0) return processed == 0 ? 2u : (seen_newline ? 1u : 0u);    /* :841-860 */
        return pw_uniform(in[pos - 1] == '\n' ? 1u : 0u);
    }

    /* the list of a search that (re)starts at `pos`: closure of instruction 0 with an all -1 vector */
    __device__ __forceinline__ void seed(int64_t pos)
    {
        const sre_pwave_list_t L = lists[0 * SRE_PWAVE_NCTX + ctx_at(pos)];
        const bool             valid = lane < L.len;
        sre_pwave_entry_t      e;
        e.tid = 0;
        e.saves = 0;
        if (valid) e = entry(L.off + lane);
        if (valid) tidvb[(cur) * 64u + lane] = e.tid;
        lab[(cur) * 64u + lane] = lane;
        const int64_t at = processed + pos;
        for (uint32_t s = 0; s < nslots; s++) {
            if (valid) capsb[(size_t) (cur) * nslots_x64() + s * 64 + lane] = ((e.saves >> s) & 1) ? at : (int64_t) -1;
        }
        n = pw_uniform(L.len);          /* (every lane read the same entry: tell the compiler) */
        if (pw_uniform(L.sss)) sss = 1;
    }

    /* sre_vm_pike.c:992-1061, 64 bytes at a time */
    __device__ __forceinline__ int64_t find_first_byte(int64_t pos, int64_t last) const
    {
        while (pos < last) {
            const int64_t  p = pos + lane;
            bool           hit = false;
            if (p < last) {
                const uint32_t c = in[p];
                hit = (lead[c >> 5] >> (c & 31)) & 1;
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
            if (m) return pos + __builtin_ctzll(m);
            pos += 64;
        }
        return last;
    }

    /* first position >= pos whose byte is not in the stable set (or `last`): eight 64-byte loads in flight */
    __device__ __forceinline__ int64_t skip_stable(int64_t pos, int64_t last) const
    {
        while (pos < last) {
            uint32_t c[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int64_t p = pos + k * 64 + lane;
                c[k] = p < last ? (uint32_t) in[p] : 256u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool     stop = c[k] > 255u || !((stab[(c[k] >> 5) & 7u] >> (c[k] & 31)) & 1u);
                const uint64_t m = __builtin_amdgcn_ballot_w64(stop);
                if (m) return pos + k * 64 + __builtin_ctzll(m);
            }
            pos += 512;
        }
        return last;
    }

    /* after the step of an iteration: its byte joins the set, or the set is dropped */
    __device__ __forceinline__ void learn(bool plain, uint32_t c, uint32_t x_in)
    {
        if (plain && step_stable && sss == x_in) {
            if (!stab_on || stab_x != x_in) {
                if (lane < 8) stab[lane] = 0;
                stab_x = x_in;
                stab_on = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == 0) stab[c >> 5] |= 1u << (c & 31);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            stab_on = 0;
        }
    }

    /* the input byte at `pos` (< last): the wave loads 64 bytes at a time, lane l byte l, and a
     * step reads its byte out of a lane — a load per byte was a third of a thin list's step */
    __device__ __forceinline__ uint32_t byte_at(int64_t pos, int64_t last)
    {
        if (pos < win_base || pos >= win_base + 64) {
            win_base = pos;
            const int64_t p = pos + lane;
            win = p < last ? (uint32_t) in[p] : 0u;
        }
        return pw_readlane(win, (uint32_t) (pos - win_base));
    }
    __device__ inline void window_reset() { win_base = -((int64_t) 1 << 62); }

    /* slot 1 of the match just recorded (last_matched_pos, :530-532, :895) */
    __device__ __forceinline__ int64_t matched_end() const
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int64_t m1 = matched[1];
        return (int64_t) (((uint64_t) pw_uniform((uint32_t) ((uint64_t) m1 >> 32)) << 32) | pw_uniform((uint32_t) (uint64_t) m1));
    }

    /* one byte step at `sp` (:312-581); returns whether a MATCH was reached */
    __device__ __forceinline__ bool step(int64_t sp, int64_t last)
    {
        const bool     at_end = sp == last;
        const uint32_t c = at_end ? 0u : byte_at(sp, last);
        const uint32_t nxt = cur ^ 1u;
        const int64_t  pos1 = processed + sp + 1;
        uint32_t       t = 0, is_m = 0, acc = 0;
        uint32_t       l_off = 0, l_len = 0, l_done = 0, l_sss = 0;
        if (lane < n) {
            t = tidvb[(cur) * 64u + lane];
            is_m = tid_match[t];
            if (!at_end && !is_m) {
                acc = (accw[t * 8 + (c >> 5)] >> (c & 31)) & 1u;
                if (acc) {
                    /* ^ behind the byte goes by the byte (:851-860); \A never holds there */
                    const sre_pwave_list_t L = lists[(uint32_t) tid_list[t] * SRE_PWAVE_NCTX + (c == '\n' ? 1u : 0u)];
                    l_off = L.off;
                    l_len = L.len;
                    l_done = L.done;
                    l_sss = L.sss;
                }
            }
        }
        const uint64_t mm = __builtin_amdgcn_ballot_w64(is_m != 0);
        const uint32_t m = mm ? (uint32_t) __builtin_ctzll(mm) : 64u;
        uint64_t       src = __builtin_amdgcn_ballot_w64(acc != 0);
        if (m < 64) src &= (1ull << m) - 1;                 /* a listed MATCH cuts what is behind it (:530-553) */
        const uint64_t dm = __builtin_amdgcn_ballot_w64(l_done != 0) & src;
        const uint32_t d = dm ? (uint32_t) __builtin_ctzll(dm) : 64u;
        if (d < 63) src &= (2ull << d) - 1;                 /* SRE_DONE ends the step behind its source (:895-898) */
        /* ---- the closures of all consuming threads at once.  Candidate c = (source, position in the
         * source's static list), numbered in the reference's order (sources by priority, then closure
         * order); lane c takes candidate c.  The scalar loop below hands the lanes to the sources with
         * two selects per source and no memory access; then ONE round for all candidates: load the
         * entry, claim the target thread with an LDS max of (step stamp, 63 - c) — the FIRST candidate
         * of a thread wins, which is what the generation tags do (:770, :792) —, read the claim back,
         * rank the winners by a prefix count, hand the owner's capture column over.  More than 64
         * candidates: batches of whole sources; a later batch first drops what the step has listed. */
        if (stamp_cur >= (1u << 26) - 2u) {
            stamp[lane] = 0;
            stamp_cur = 0;
            lab[(cur) * 64u + lane] = lane;         /* (fresh labels are (stamp << 6) | rank: start over, all distinct) */
        }
        stamp_cur++;
        uint32_t nn = 0;
        bool     done = false, first_batch = true, moved = false;
        uint32_t d_off = 0, d_len = 0;
        for (uint64_t rem = src; rem;) {
            uint32_t owner = 0, kbase = 0, loff = 0, total = 0;
            while (rem) {
                const uint32_t i = (uint32_t) __builtin_ctzll(rem);
                const uint32_t len = pw_readlane(l_len, i);
                if (total + len > 64u) break;
                const uint32_t off = pw_readlane(l_off, i);
                const bool     ge = lane >= total;
                owner = ge ? i : owner;
                kbase = ge ? total : kbase;
                loff = ge ? off : loff;
                total += len;
                if (pw_readlane(l_sss, i)) sss = 1;
                if (i == d) {
                    d_off = off;
                    d_len = len;
                }
                rem &= rem - 1;
            }
            const bool        valid = lane < total;
            sre_pwave_entry_t e;
            e.tid = 0;
            e.saves = 0;
            if (valid) e = entry(loff + (lane - kbase));
            bool keep = valid;
            if (!first_batch) keep = keep && (stamp[e.tid] >> 6) != stamp_cur;
            const uint32_t key = (stamp_cur << 6) | (63u - lane);
            if (keep) __hip_atomic_fetch_max(&stamp[e.tid], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            keep = keep && stamp[e.tid] == key;
            const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
            const uint32_t rank = nn + pw_lane_rank(km);
            if (keep) tidvb[(nxt) * 64u + rank] = e.tid;
            /* stable runs: does every place keep its thread and its column — nothing saved on the way, and the
             * parent's column known to equal the one the place held (labels)? */
            const uint32_t lo = lab[(cur) * 64u + owner];
            if (keep) lab[(nxt) * 64u + rank] = e.saves ? ((stamp_cur << 6) | rank) : lo;
            moved = moved || __builtin_amdgcn_ballot_w64(keep && (e.saves != 0 || tidvb[(cur) * 64u + rank] != e.tid || lab[(cur) * 64u + rank] != lo)) != 0;
            /* the owner's capture vector, slot by slot */
            for (uint32_t s = 0; s < nslots; s++) {
                const int64_t v = capsb[(size_t) (cur) * nslots_x64() + s * 64 + owner];
                if (keep) capsb[(size_t) (nxt) * nslots_x64() + s * 64 + rank] = ((e.saves >> s) & 1) ? pos1 : v;
            }
            nn += (uint32_t) __builtin_popcountll(km);
            first_batch = false;
        }
        if (d < 64) {
            /* the closure of source d reached MATCH behind its listed targets: the match (:895-898) */
            const sre_pwave_entry_t me = entry(d_off + d_len);
            if (lane < nslots) matched[lane] = ((me.saves >> lane) & 1) ? pos1 : capsb[(size_t) (cur) * nslots_x64() + lane * 64 + d];
            matched_id = (int64_t) pw_uniform(tid_match[me.tid]) - 1;
            done = true;
        }
        if (!done && m < 64) {
            /* the listed MATCH thread is reached (:530-553) */
            if (lane < nslots) matched[lane] = capsb[(size_t) (cur) * nslots_x64() + lane * 64 + m];
            matched_id = (int64_t) pw_readlane(is_m, m) - 1;
            done = true;
        }
        if (done) has_matched = 1;
        step_stable = !at_end && !done && !moved && nn == n;
        cur = nxt;
        n = nn;
        return done;
    }

    /* one whole-buffer exec() of a fresh (possibly re-armed) context with eof, picked up at `start`
     * (sre_hip_vm.hip Pike::exec) */
    __device__ __forceinline__ int64_t exec(int64_t size, int64_t start, bool start_is_skip_target, int64_t *ov, uint32_t ovec_slots)
    {
        const int64_t last = size;
        int64_t       sp = 0;
        bool          no_check_once = false, skip_ran_out = false;
        int64_t       last_matched_pos = -1;
        window_reset();
        if (ctx_eof) return -1;                                 /* :165-168 */
        if (empty_capture) {                                    /* :179-196 */
            empty_capture = 0;
            if (size == 0) {
                ctx_eof = 1;
                return PW_RC_DECLINED;
            }
            sp = 1;
        }
        cur = 0;
        n = 0;
        sss = 0;
        has_matched = 0;
        poisoned = 0;
        stamp_cur = 0;
        stamp[lane] = 0;
        /* :202-233 */
        seed(sp);
        initial_count = n;
        if (lane + 1 < n) initial[lane] = tidvb[(cur) * 64u + lane];      /* all but its last thread (:218-229) */
        if (start > sp) {
            sp = start;
            if (nleading && start_is_skip_target) {
                sp = find_first_byte(sp, last);
                no_check_once = true;
                if (sp == last) skip_ran_out = true;            /* :304-306 */
            }
            seed(sp);
        }

        stab_on = 0;
        for (; !skip_ran_out && sp <= last; sp++) {             /* :235 (eof) */
            if (n == 0) break;
            bool     plain = true;
            uint32_t x_in = sss;
            if (no_check_once) {
                no_check_once = false;
                plain = false;
            } else {
                if (stab_on && sss == stab_x && sp < last) sp = skip_stable(sp, last);     /* a stable run */
                x_in = sss;
                if (nleading && sss) {                       /* :256-309 */
                    sss = 0;
                    bool same = (sp != last) && (n == initial_count);
                    if (same) {
                        const bool diff = lane + 1 < n && tidvb[(cur) * 64u + lane] != initial[lane];
                        same = __builtin_amdgcn_ballot_w64(diff) == 0;
                    }
                    if (same) {
                        const int64_t p = find_first_byte(sp, last);
                        if (p > sp) {
                            sp = p;
                            seed(sp);
                            plain = false;
                            if (sp == last) break;
                        }
                    }
                }
            }
            const uint32_t c = sp < last ? byte_at(sp, last) : 0u;
            if (step(sp, last)) last_matched_pos = matched_end();
            learn(plain && sp < last, c, x_in);
            if (sp == last) break;
        }

        if (last_matched_pos >= 0) {                            /* :586-601 (seen_word: only look-ahead programs read it) */
            const int64_t p = last_matched_pos - processed;
            if (p > 0) seen_newline = pw_uniform(in[p - 1] == '\n' ? 1u : 0u);
        }
        if (has_matched) {                                      /* :607-636, eof */
            if (matched_id >= (int64_t) nregexes) return -1;
            const uint32_t *ncaps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(W) + W->multi_ncaps_off);
            uint32_t        ofs = 0;
            for (int64_t r = 0; r < matched_id; r++) ofs += ncaps[r] + 1;
            ofs *= 2;
            const uint32_t cnt = 2u * (ncaps[matched_id] + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t k = lane; k < ovec_slots; k += 64) ov[k] = k < cnt ? matched[ofs + k] : (int64_t) -1;
            if (n > 0) {
                poisoned = 1;                                   /* :616-622 */
                ctx_eof = 1;
            }
            /* :624-628 re-arm for the next search on the same context */
            const int64_t m0 = matched[ofs], m1 = matched[ofs + 1];
            processed = (int64_t) (((uint64_t) pw_uniform((uint32_t) ((uint64_t) m1 >> 32)) << 32) | pw_uniform((uint32_t) (uint64_t) m1));
            empty_capture = pw_uniform(m0 == m1 ? 1u : 0u);
            return matched_id;
        }
        ctx_eof = 1;
        return PW_RC_DECLINED;                                  /* :660-666 */
    }

    /*
     * One exec() of a context fed in CHUNKS (sre_vm_pike.c:148-689 with or without eof), as
     * sre_hip_vm.hip Pike::exec takes it with one lane: the list and its capture columns stay in the
     * context between calls; without a decision the call answers SRE_AGAIN with the range a match
     * could still span (:692-735) and the pending match (:640-658).
     */
    __device__ __forceinline__ int64_t exec_chunk(int64_t size, bool eof, bool want_pending, int64_t *ov, uint32_t ovec_slots,
                                  int64_t *has_pending, int64_t *pending, int64_t *consumed)
    {
        const int64_t last = size;
        int64_t       sp = 0, last_matched_pos = -1;
        window_reset();
        *has_pending = 0;
        *consumed = 0;
        if (ctx_eof) return -1;                                 /* :165-168 */
        if (empty_capture) {                                    /* :179-196 */
            empty_capture = 0;
            if (size == 0) {
                if (eof) {
                    ctx_eof = 1;
                    return PW_RC_DECLINED;
                }
                return -2;
            }
            sp = 1;
        }
        stamp_cur = 0;
        stamp[lane] = 0;
        if (first_buf) {                                        /* :202-233 */
            first_buf = 0;
            sss = 0;
            seed(sp);
            initial_count = n;
            if (lane + 1 < n) initial[lane] = tidvb[(cur) * 64u + lane];
        }
        stab_on = 0;
        for (; sp < last || (eof && sp == last); sp++) {        /* :235 */
            if (n == 0) break;
            bool plain = true;
            if (stab_on && sss == stab_x && sp < last) {        /* a stable run */
                sp = skip_stable(sp, last);
                if (sp == last && !eof) break;
            }
            const uint32_t x_in = sss;
            if (nleading && sss) {                           /* :256-309 */
                sss = 0;
                bool same = (sp != last) && (n == initial_count);
                if (same) {
                    const bool diff = lane + 1 < n && tidvb[(cur) * 64u + lane] != initial[lane];
                    same = __builtin_amdgcn_ballot_w64(diff) == 0;
                }
                if (same) {
                    const int64_t p = find_first_byte(sp, last);
                    if (p > sp) {
                        sp = p;
                        seed(sp);
                        plain = false;
                        if (sp == last) break;
                    }
                }
            }
            const uint32_t c = sp < last ? byte_at(sp, last) : 0u;
            if (step(sp, last)) last_matched_pos = matched_end();
            learn(plain && sp < last, c, x_in);
            if (sp == last) break;
        }
        *consumed = sp;

        if (last_matched_pos >= 0) {                            /* :586-601 */
            const int64_t p = last_matched_pos - processed;
            if (p > 0) {
                const uint32_t b = pw_uniform((uint32_t) in[p - 1]);
                seen_newline = b == '\n';
                seen_word = (b - '0' < 10u) || (b - 'A' < 26u) || (b - 'a' < 26u) || b == '_';
            }
        }
        const uint32_t *ncaps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(W) + W->multi_ncaps_off);
        if (has_matched) {                                      /* :607-658 */
            if (matched_id >= (int64_t) nregexes) return -1;
            uint32_t ofs = 0;
            for (int64_t r = 0; r < matched_id; r++) ofs += ncaps[r] + 1;
            ofs *= 2;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int64_t m0 = matched[ofs], m1r = matched[ofs + 1];
            const int64_t m1 = (int64_t) (((uint64_t) pw_uniform((uint32_t) ((uint64_t) m1r >> 32)) << 32)
                                          | pw_uniform((uint32_t) (uint64_t) m1r));
            if (eof || n == 0) {
                const uint32_t cnt = 2u * (ncaps[matched_id] + 1);
                for (uint32_t k = lane; k < ovec_slots; k += 64) ov[k] = k < cnt ? matched[ofs + k] : (int64_t) -1;
                if (n > 0) {
                    n = 0;                                      /* :616-622 */
                    ctx_eof = 1;
                }
                processed = m1;                                 /* :624-628 */
                empty_capture = pw_uniform(m0 == m1r ? 1u : 0u);
                has_matched = 0;
                first_buf = 1;
                return matched_id;
            }
            if (want_pending) {
                *has_pending = 1;
                pending[0] = m0;
                pending[1] = m1;
            }
        } else if (eof) {                                       /* :660-666 */
            ctx_eof = 1;
            return PW_RC_DECLINED;
        }
        processed += sp;                                        /* :673-688 */
        if (ovec_slots >= 2) {
            /* :692-735 (end offset read without the per-regex offset, :721) */
            int64_t a0 = INT64_MAX, a1 = -1;
            if (lane < n) {
                uint32_t ofs = 0;
                for (uint32_t r = 0; r < nregexes; r++) {
                    const int64_t b = capsb[(size_t) (cur) * nslots_x64() + ofs * 64 + lane];
                    if (b != -1 && b < a0) a0 = b;
                    ofs += 2 * (ncaps[r] + 1);
                }
                a1 = capsb[(size_t) (cur) * nslots_x64() + 1 * 64 + lane];
            }
            for (int d = 32; d > 0; d >>= 1) {
                const int64_t o0 = __shfl_xor(a0, d), o1 = __shfl_xor(a1, d);
                a0 = o0 < a0 ? o0 : a0;
                a1 = o1 > a1 ? o1 : a1;
            }
            if (lane == 0) {
                ov[0] = a0 == INT64_MAX ? (int64_t) -1 : a0;
                ov[1] = a1;
            }
        }
        return -2;
    }
};

}  // namespace

/* LDS a window wave needs for a program */
static size_t
pwave_lds_bytes(const sre_pwave_hdr_t *h, bool *ents_in_lds)
{
    size_t b = 0;
    b += 2 * (size_t) h->nslots * 64 * 8;       /* caps */
    b += 64 * 8;                                /* matched */
    b += 16;                                    /* a chunk that travels in the request (sre_k_pike_exec_wave) */
    b += 32;                                    /* leading-byte map */
    b += 32 + 2 * 64 * 4;                       /* stable-run byte set, column labels */
    b += 2 * 64 * 2 + 64 * 4 + 64 * 2;          /* tidv, stamp, initial */
    b += 64 * 8 * 4 + 64 * 2 * 2;               /* accept words, tid_list, tid_match */
    b += (size_t) h->nlists * SRE_PWAVE_NCTX * sizeof(sre_pwave_list_t);
    const size_t eb = ((size_t) h->nentries + 1) * sizeof(sre_pwave_entry_t);
    *ents_in_lds = b + eb <= 60 * 1024;
    if (*ents_in_lds) b += eb;
    return (b + 15) & ~(size_t) 15;
}

/* carve the wave's LDS and copy the program's tables into it */
template <bool ENTS_LDS>
__device__ __forceinline__ void
pwave_setup(PikeWave<ENTS_LDS> &vm, const sre_pwave_hdr_t *Wg, uint8_t *lds, uint32_t lane)
{
    const uint32_t nslots = Wg->nslots;
    uint8_t       *p = lds;
    vm.capsb = reinterpret_cast<int64_t *>(p);      p += 2 * (size_t) nslots * 64 * 8;
    vm.matched = reinterpret_cast<int64_t *>(p);    p += 64 * 8;
    vm.inl = p;                                     p += 16;
    uint32_t *lead = reinterpret_cast<uint32_t *>(p);   p += 32;
    if (lane < 8) lead[lane] = Wg->lead[lane];
    vm.lead = lead;
    vm.stab = reinterpret_cast<uint32_t *>(p);      p += 32;
    vm.lab = reinterpret_cast<uint32_t *>(p);       p += 2 * 64 * 4;
    vm.stab_on = 0;
    vm.stab_x = 0;
    vm.step_stable = 0;
    uint32_t *accw = reinterpret_cast<uint32_t *>(p);   p += 64 * 8 * 4;
    vm.stamp = reinterpret_cast<uint32_t *>(p);     p += 64 * 4;
    vm.tidvb = reinterpret_cast<uint16_t *>(p);     p += 2 * 64 * 2;
    vm.initial = reinterpret_cast<uint16_t *>(p);   p += 64 * 2;
    uint16_t *tid_list = reinterpret_cast<uint16_t *>(p);   p += 64 * 2;
    uint16_t *tid_match = reinterpret_cast<uint16_t *>(p);  p += 64 * 2;
    sre_pwave_list_t *lists = reinterpret_cast<sre_pwave_list_t *>(p);
    p += (size_t) Wg->nlists * SRE_PWAVE_NCTX * sizeof(sre_pwave_list_t);
    p = lds + (((size_t) (p - lds) + 15) & ~(size_t) 15);
    const uint8_t *wb = reinterpret_cast<const uint8_t *>(Wg);
    for (uint32_t k = lane; k < 64 * 8; k += 64) accw[k] = Wg->accept[k >> 3][k & 7];
    tid_list[lane] = Wg->tid_list[lane];
    tid_match[lane] = Wg->tid_match[lane];
    {
        const sre_pwave_list_t *gl = reinterpret_cast<const sre_pwave_list_t *>(wb + Wg->off_lists);
        for (uint32_t k = lane; k < Wg->nlists * SRE_PWAVE_NCTX; k += 64) lists[k] = gl[k];
    }
    const sre_pwave_entry_t *ents = reinterpret_cast<const sre_pwave_entry_t *>(wb + Wg->off_entries);
    vm.ents_lds = ~0u;
    if (ENTS_LDS) {
        sre_pwave_entry_t *le = reinterpret_cast<sre_pwave_entry_t *>(p);
        for (uint32_t k = lane; k < Wg->nentries + 1; k += 64) le[k] = ents[k];
        vm.ents_lds = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint8_t *) p;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    vm.W = Wg;
    vm.nleading = pw_uniform(Wg->nleading);
    vm.nregexes = pw_uniform(Wg->nregexes);
    vm.accw = accw;
    vm.lists = lists;
    vm.ents = ents;
    vm.tid_list = tid_list;
    vm.tid_match = tid_match;
    vm.nslots = nslots;
    vm.lane = lane;
    vm.processed = 0;
    vm.seen_newline = 0;
    vm.empty_capture = 0;
    vm.ctx_eof = 0;
    vm.first_buf = 1;
    vm.seen_word = 0;
}

/*
 * ENGINE_VM, Pike: whole streams, one wavefront each — first match, or the find-all iteration of the
 * reference's caller on one re-armed context (exec from the previous match's end until SRE_DECLINED,
 * sre_vm_pike.c:179-196, :586-636), as sre_hip_vm.hip sre_k_pike_scan does with one lane per stream.
 */
template <bool ENTS_LDS>
__global__ __launch_bounds__(64) void
sre_k_pike_scan_wave(const sre_pwave_hdr_t *__restrict__ Wg, const uint8_t *const *__restrict__ streams,
                     const uint64_t *__restrict__ lens, uint32_t nstreams, int64_t *__restrict__ records,
                     uint32_t ovec_slots, int mode)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t i = blockIdx.x;
    if (i >= nstreams) return;
    const uint32_t lane = threadIdx.x;
    PikeWave<ENTS_LDS> vm;
    pwave_setup(vm, Wg, lds, lane);
    int64_t       *rec = records + (size_t) i * (2 + ovec_slots);
    const uint8_t *s = streams[i];
    const uint64_t n = lens[i];
    uint64_t       off = 0;
    int64_t        count = 0, rc, last_rc = PW_RC_DECLINED;
    for (uint32_t k = lane; k < ovec_slots; k += 64) rec[2 + k] = -1;
    for (;;) {
        vm.in = s + off;
        rc = vm.exec((int64_t) (n - off), 0, false, rec + 2, ovec_slots);
        if (rc < 0) break;
        count++;
        last_rc = rc;
        if (mode != 2) break;
        off = (uint64_t) vm.processed;      /* == ovector[1] of this match */
    }
    /* the final SRE_DECLINED leaves the ovector of the last match in place (:660-666 writes nothing) */
    if (lane == 0) {
        rec[0] = rc == -1 ? rc : (count > 0 ? last_rc : rc);
        rec[1] = count;
    }
}

template <bool ENTS_LDS>
__global__ __launch_bounds__(64) void
sre_k_pike_window_wave(const sre_pwave_hdr_t *__restrict__ Wg, const uint8_t *const *__restrict__ streams,
                       const uint64_t *__restrict__ lens, uint32_t nstreams, int64_t *__restrict__ records,
                       uint32_t ovec_slots, sre_nfa_window_t *__restrict__ win, const int64_t *__restrict__ lo,
                       const sre_nfa_count_req_t *__restrict__ creq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t i = blockIdx.x;
    if (i >= nstreams) return;
    if (lo != nullptr && lo[i] < 0) return;     /* settled in an earlier round */
    if (!win[i].done || win[i].ev_pos < 0) return;
    const uint32_t lane = threadIdx.x;

    PikeWave<ENTS_LDS> vm;
    pwave_setup(vm, Wg, lds, lane);
    vm.in = streams[i];
    vm.processed = 0;
    vm.seen_newline = 0;
    vm.empty_capture = 0;
    vm.ctx_eof = 0;

    int64_t *rec = records + (size_t) i * (2 + ovec_slots);
    uint64_t len = lens[i];
    int64_t  start = win[i].clean_pos;
    if (creq != nullptr) {
        /* a search of a find-all iteration: the reference's re-armed context */
        vm.in = creq[i].vptr;
        len = creq[i].vlen;
        start += creq[i].start_add;
        vm.processed = creq[i].processed;
        vm.seen_newline = (creq[i].preset_flags & SRE_PRESET_SEEN_NEWLINE) ? 1u : 0u;
    }
    const int64_t rc = vm.exec((int64_t) len, start, (win[i].clean_mode & 1) != 0, rec + 2, ovec_slots);
    if (rc < 0) {
        for (uint32_t k = lane; k < ovec_slots; k += 64) rec[2 + k] = -1;
    }
    if (lane == 0) {
        rec[0] = rc;
        rec[1] = rc >= 0 ? 1 : 0;
        int32_t cm = win[i].clean_mode;
        if (rc >= 0 && vm.poisoned) cm |= SRE_NFA_WINDOW_POISONED;
        if (rc >= 0 && creq != nullptr && ovec_slots >= 2) {
            /* what the next search's ^ goes by (seen_newline, :586-601) */
            const int64_t e = vm.matched[1] - creq[i].processed;
            /* (slot 1 of the internal vector: the end of a match of regex 0; other regexes' matches
             * leave last_matched_pos unset and the flag unchanged — the host keeps its own) */
            if (e > 0 && vm.in[e - 1] == '\n') cm |= SRE_NFA_MATCH_AFTER_NL;
        }
        win[i].clean_mode = cm;
    }
}

extern "C" hipError_t
sre_launch_pike_window_wave(const void *d_wave_v, const void *h_wave_v, const void *const *d_streams,
                            const uint64_t *d_lens, uint32_t nstreams, int64_t *d_records, uint32_t ovec_slots,
                            sre_nfa_window_t *d_win, const int64_t *d_lo, const sre_nfa_count_req_t *d_creq,
                            hipStream_t stream)
{
    const sre_pwave_hdr_t *d_wave = static_cast<const sre_pwave_hdr_t *>(d_wave_v);
    const sre_pwave_hdr_t *h_wave = static_cast<const sre_pwave_hdr_t *>(h_wave_v);
    bool         in_lds = false;
    const size_t bytes = pwave_lds_bytes(h_wave, &in_lds);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(in_lds ? reinterpret_cast<const void *>(sre_k_pike_window_wave<true>) : reinterpret_cast<const void *>(sre_k_pike_window_wave<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
        if (e != hipSuccess) return e;
    }
    if (in_lds) {
        hipLaunchKernelGGL(sre_k_pike_window_wave<true>, dim3(nstreams), dim3(64), bytes, stream, d_wave,
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams, d_records, ovec_slots,
                           d_win, d_lo, d_creq);
    } else {
        hipLaunchKernelGGL(sre_k_pike_window_wave<false>, dim3(nstreams), dim3(64), bytes, stream, d_wave,
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams, d_records, ovec_slots,
                           d_win, d_lo, d_creq);
    }
    return hipGetLastError();
}

extern "C" hipError_t
sre_launch_pike_scan_wave(const void *d_wave_v, const void *h_wave_v, int mode, const void *const *d_streams,
                          const uint64_t *d_lens, uint32_t nstreams, int64_t *d_records, uint32_t ovec_slots,
                          hipStream_t stream)
{
    const sre_pwave_hdr_t *d_wave = static_cast<const sre_pwave_hdr_t *>(d_wave_v);
    const sre_pwave_hdr_t *h_wave = static_cast<const sre_pwave_hdr_t *>(h_wave_v);
    bool         in_lds = false;
    const size_t bytes = pwave_lds_bytes(h_wave, &in_lds);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(in_lds ? reinterpret_cast<const void *>(sre_k_pike_scan_wave<true>) : reinterpret_cast<const void *>(sre_k_pike_scan_wave<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
        if (e != hipSuccess) return e;
    }
    if (in_lds) {
        hipLaunchKernelGGL(sre_k_pike_scan_wave<true>, dim3(nstreams), dim3(64), bytes, stream, d_wave,
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams, d_records, ovec_slots, mode);
    } else {
        hipLaunchKernelGGL(sre_k_pike_scan_wave<false>, dim3(nstreams), dim3(64), bytes, stream, d_wave,
                           reinterpret_cast<const uint8_t *const *>(d_streams), d_lens, nstreams, d_records, ovec_slots, mode);
    }
    return hipGetLastError();
}

/*
 * The compat API's streaming VM (sre_vm_pike_exec fed chunk by chunk, sre_vm_api.cpp): one request =
 * one exec() on one context, taken by a wavefront.  The context between calls (device memory,
 * zero-filled == fresh): the header below, the current list and its capture columns.
 */
#define SRE_PWAVE_CTX_MAGIC 0x50575631u

struct sre_pwave_ctx_t {
    uint32_t magic, n, sss, initial_count;
    uint32_t has_matched, first_buf, empty_capture, ctx_eof;
    uint32_t seen_newline, seen_word, pad[2];
    int64_t  processed, matched_id;
    uint16_t tidv[64], initial[64];
    int64_t  matched[64];
    /* int64_t caps[nslots][64] follows */
};

extern "C" size_t
sre_pwave_ctx_bytes(const void *h_wave)
{
    return sizeof(sre_pwave_ctx_t) + (size_t) static_cast<const sre_pwave_hdr_t *>(h_wave)->nslots * 64 * 8;
}

template <bool ENTS_LDS>
__global__ __launch_bounds__(64) void
sre_k_pike_exec_wave(const sre_pwave_hdr_t *__restrict__ Wg, const sre_dev_req_t *__restrict__ reqs, uint32_t nreqs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t i = blockIdx.x;
    if (i >= nreqs) return;
    const sre_dev_req_t rq = reqs[i];
    const uint32_t      lane = threadIdx.x;
    PikeWave<ENTS_LDS>  vm;
    pwave_setup(vm, Wg, lds, lane);
    const uint32_t   nslots = vm.nslots;
    sre_pwave_ctx_t *cx = static_cast<sre_pwave_ctx_t *>(rq.ctx);
    int64_t         *ccaps = reinterpret_cast<int64_t *>(cx + 1);
    vm.cur = 0;
    vm.n = 0;
    vm.sss = 0;
    vm.initial_count = 0;
    vm.has_matched = 0;
    vm.poisoned = 0;
    vm.matched_id = 0;
    if (rq.fresh || pw_uniform(cx->magic) != SRE_PWAVE_CTX_MAGIC) {
        /* fresh context (sre_vm_pike.c:94-145); earlier searches may have run on the scanner */
        if (rq.preset_valid) {
            vm.processed = rq.preset_processed;
            vm.empty_capture = (rq.preset_flags & SRE_PRESET_EMPTY_CAPTURE) ? 1u : 0u;
            vm.seen_newline = (rq.preset_flags & SRE_PRESET_SEEN_NEWLINE) ? 1u : 0u;
            vm.seen_word = (rq.preset_flags & SRE_PRESET_SEEN_WORD) ? 1u : 0u;
            vm.ctx_eof = (rq.preset_flags & SRE_PRESET_EOF) ? 1u : 0u;
        }
    } else {
        vm.n = pw_uniform(cx->n);
        vm.sss = pw_uniform(cx->sss);
        vm.initial_count = pw_uniform(cx->initial_count);
        vm.has_matched = pw_uniform(cx->has_matched);
        vm.first_buf = pw_uniform(cx->first_buf);
        vm.empty_capture = pw_uniform(cx->empty_capture);
        vm.ctx_eof = pw_uniform(cx->ctx_eof);
        vm.seen_newline = pw_uniform(cx->seen_newline);
        vm.seen_word = pw_uniform(cx->seen_word);
        vm.processed = cx->processed;
        vm.matched_id = cx->matched_id;
        vm.tidvb[(0) * 64u + lane] = cx->tidv[lane];
        vm.lab[(0) * 64u + lane] = lane;
        vm.initial[lane] = cx->initial[lane];
        vm.matched[lane] = cx->matched[lane];
        if (lane < vm.n) {
            for (uint32_t s = 0; s < nslots; s++) vm.capsb[(size_t) (0) * vm.nslots_x64() + s * 64 + lane] = ccaps[s * 64 + lane];
        }
    }
    /* a chunk of up to 8 bytes travels in the request, one of up to SRE_SMALL_INPUT in pinned host
     * memory next to it (staged into LDS, all loads in flight at once) */
    __shared__ __attribute__((aligned(16))) uint8_t sh_small_in[SRE_SMALL_INPUT];
    if (rq.input == nullptr) {
        if (lane < 8) vm.inl[lane] = (uint8_t) (rq.inline_bytes >> (8 * lane));
        vm.in = vm.inl;
    } else if (rq.input_pinned) {
        for (uint32_t k = lane * 16; k < rq.size; k += 64 * 16) {
            *reinterpret_cast<uint4 *>(sh_small_in + k) = *reinterpret_cast<const uint4 *>(rq.input + k);
        }
        vm.in = sh_small_in;
    } else {
        vm.in = rq.input;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    sre_dev_result_t *res = static_cast<sre_dev_result_t *>(rq.result);
    int64_t          *ov = reinterpret_cast<int64_t *>(res + 1);
    int64_t           has_pending = 0, pending[2] = {0, 0}, consumed = 0;
    const int64_t     rc = vm.exec_chunk((int64_t) rq.size, rq.eof != 0, rq.want_pending != 0, ov, (uint32_t) rq.ovec_slots,
                                         &has_pending, pending, &consumed);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    /* the context for the next call */
    cx->tidv[lane] = vm.tidvb[(vm.cur) * 64u + lane];
    cx->initial[lane] = vm.initial[lane];
    cx->matched[lane] = vm.matched[lane];
    if (lane < vm.n) {
        for (uint32_t s = 0; s < nslots; s++) ccaps[s * 64 + lane] = vm.capsb[(size_t) (vm.cur) * vm.nslots_x64() + s * 64 + lane];
    }
    if (lane == 0) {
        cx->magic = SRE_PWAVE_CTX_MAGIC;
        cx->n = vm.n;
        cx->sss = vm.sss;
        cx->initial_count = vm.initial_count;
        cx->has_matched = vm.has_matched;
        cx->first_buf = vm.first_buf;
        cx->empty_capture = vm.empty_capture;
        cx->ctx_eof = vm.ctx_eof;
        cx->seen_newline = vm.seen_newline;
        cx->seen_word = vm.seen_word;
        cx->processed = vm.processed;
        cx->matched_id = vm.matched_id;
        res->has_pending = has_pending;
        res->pending[0] = pending[0];
        res->pending[1] = pending[1];
        res->consumed = consumed;
        /* what the context holds between two searches (sre_hip_vm.hip sre_k_pike_exec) */
        res->pad[0] = 16 | (vm.empty_capture ? SRE_PRESET_EMPTY_CAPTURE : 0) | (vm.seen_newline ? SRE_PRESET_SEEN_NEWLINE : 0)
                      | (vm.seen_word ? SRE_PRESET_SEEN_WORD : 0) | (vm.ctx_eof ? SRE_PRESET_EOF : 0);
        res->pad[1] = vm.processed;
        /* the host watches res->rc (device_stream_exec): written last, behind a system-scope fence */
        __threadfence_system();
        *reinterpret_cast<volatile int64_t *>(&res->rc) = rc;
    }
}

extern "C" hipError_t
sre_launch_pike_exec_wave(const void *d_wave_v, const void *h_wave_v, const sre_dev_req_t *d_reqs, uint32_t nreqs,
                          hipStream_t stream)
{
    const sre_pwave_hdr_t *d_wave = static_cast<const sre_pwave_hdr_t *>(d_wave_v);
    const sre_pwave_hdr_t *h_wave = static_cast<const sre_pwave_hdr_t *>(h_wave_v);
    bool         in_lds = false;
    const size_t bytes = pwave_lds_bytes(h_wave, &in_lds);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(in_lds ? reinterpret_cast<const void *>(sre_k_pike_exec_wave<true>) : reinterpret_cast<const void *>(sre_k_pike_exec_wave<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
        if (e != hipSuccess) return e;
    }
    if (in_lds) {
        hipLaunchKernelGGL(sre_k_pike_exec_wave<true>, dim3(nreqs), dim3(64), bytes, stream, d_wave, d_reqs, nreqs);
    } else {
        hipLaunchKernelGGL(sre_k_pike_exec_wave<false>, dim3(nreqs), dim3(64), bytes, stream, d_wave, d_reqs, nreqs);
    }
    return hipGetLastError();
}

/* does a wave's state for this program fit the LDS of a workgroup? */
extern "C" int
sre_pwave_fits(const void *h_wave)
{
    bool in_lds;
    return pwave_lds_bytes(static_cast<const sre_pwave_hdr_t *>(h_wave), &in_lds) <= 96 * 1024;
}
