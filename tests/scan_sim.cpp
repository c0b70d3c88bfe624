/*
 * tests/scan_sim.cpp — TEST-ONLY host model of ONE LANE of the table-driven scanner in COUNT mode
 * (sregex_amd/csrc/sre_hip_scan.hip: the round loop, slow_run, note_span / fast_span_done / settle,
 * resolve_fast_span) over a whole stream, on the fast table sre_scan_fast_build() makes (sre_scan_fast.cpp).
 *
 * The kernel keeps the find-all iteration on its fast path by FOLDING the caller's restarts into the table:
 * matches that end with a consumed byte, that a look-ahead assertion completes (the next search reads the
 * deciding byte again), empty ones (the caller skips a byte), matches that GROW while the list lives on (FRESH
 * states: the pending match is not recorded, the lane recovers it by replaying its last span) and the list
 * dying in a FRESH state.  This model walks the same table with the same bookkeeping — spans of 16 bytes (the
 * kernel's group path) or 64 (its round path), or every byte on the exact path — and the CPU suite compares
 * count, last match and its search start with the oracle (tests/test_scan_model.py): the ALGORITHM and the
 * host-built table are pinned without a GPU.  Compiled into tests/_build/; not part of the product library.
 */
#include "sre_dfa.h"
#include "sre_scan_fast.h"
#include <sregex_hip.h>
#include <stdint.h>
#include <string.h>
#include <vector>

namespace {

enum { K_DONE = 1, K_POP = 2, K_DONE_EMPTY = 3, K_POP_FULL = 4 };      /* device event kinds (sre_hip_scan.h) */
enum : uint32_t { F_HAS_EV = 1, F_LM_VALID = 2, F_FINISHED = 4, F_ERROR = 8, F_UNRESOLVED = 16, F_SKIP_NEXT = 32,
                  F_SP_DIRTY = 128, F_PEND_LAZY = 512, F_LZ_GROUP = 1024 };

struct SpanResult {
    int64_t  sp, last_pos, last_sp, pend_pos;
    uint32_t last_state, last_sym, pend_state, pend_sym;
};

struct Lane {
    const sre_dfa_t      *d;
    sre_scan_fast_t       F;
    std::vector<uint16_t> tr2;
    const uint8_t        *data;
    int64_t               n;
    uint32_t              nsym, word_restart;
    /* Walk */
    uint32_t st = 0, fl = 0;
    uint32_t ev_kind = 0, ev_state = 0, ev_sym = 0, lm_state = 0, lm_sym = 0;
    int64_t  ev_pos = -1, ev_sp = -1, cur_sp = -1, lm_pos = -1, lm_sp = -1, count = 0, term_pos = -1;
    /* the capture walker's anchor: a state of the match's search shortly in front of its event (-1 none) */
    int64_t  anchor_pos = -1, ev_apos = -1, lm_apos = -1;
    uint32_t anchor_state = 0, ev_astate = 0, lm_astate = 0;
    /* spans */
    int64_t  fcA_pos = -1, fcB_pos = -1, fc_sp0 = -1, lz_pos = -1;
    uint32_t fcA_len = 0, fcA_s0 = 0, fcB_len = 0, fcB_s0 = 0, lz_s0 = 0;
    uint32_t span = 64;
    int64_t  fast_bytes = 0;

    bool f(uint32_t b) const { return (fl & b) != 0; }
    uint32_t variant_of(uint32_t c) const
    {
        if (c == '\n') return 1;
        if (word_restart && sre_isword(c)) return 3;
        return 2;
    }
    void complete_match()
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
    /* sre_hip_scan.hip slow_run<COUNT>, not warm */
    void slow_run(int64_t p, int64_t p_to)
    {
        if (f(F_SKIP_NEXT)) {
            fl &= ~F_SKIP_NEXT;
            p++;
        }
        while (p < p_to && !f(F_FINISHED)) {
            if (p == n && d->seen_start[st] == 2) {
                term_pos = p;
                if (f(F_HAS_EV)) {
                    complete_match();
                    fl |= F_ERROR;
                }
                fl |= F_FINISHED;
                return;
            }
            const uint32_t sym = p < n ? d->cls_map[data[p]] : d->ncls;
            const uint32_t t2 = tr2[st * nsym + sym];
            if (t2 >> 8) {
                fl |= F_HAS_EV;
                ev_kind = t2 >> 8;
                ev_state = st;
                ev_sym = sym;
                ev_pos = p;
                ev_sp = cur_sp;
                /* (strictly behind the search start: sre_hip_scan.hip slow_run) */
                if (anchor_pos >= 0 && anchor_pos <= p && p - anchor_pos <= 256 && (cur_sp < 0 || anchor_pos > cur_sp)) {
                    ev_apos = anchor_pos;
                    ev_astate = anchor_state;
                } else {
                    ev_apos = -1;
                    ev_astate = 0;
                }
            }
            st = t2 & 0xffu;
            if (st != 0) {
                p++;
                continue;
            }
            if (!f(F_HAS_EV)) {
                if (p < n) fl |= F_UNRESOLVED;
                term_pos = p;
                fl |= F_FINISHED;
                return;
            }
            const bool    pop = ev_kind == K_POP || ev_kind == K_POP_FULL;
            const bool    empty = ev_kind == K_POP || ev_kind == K_DONE_EMPTY;
            const int64_t e = pop ? ev_pos : ev_pos + 1;
            complete_match();
            if (empty) {
                if (e >= n) {
                    term_pos = p;
                    fl |= F_FINISHED;
                    return;
                }
                cur_sp = e + 1;
            } else {
                cur_sp = e;
            }
            st = d->init[variant_of(data[cur_sp - 1])];
            p = cur_sp;
            if (p > p_to && p_to <= n) fl |= F_SKIP_NEXT;
            anchor_pos = -1;        /* the span's entry state belonged to the previous search */
        }
    }
    SpanResult resolve(int64_t gpos, uint32_t len, uint32_t s0, int64_t sp0, int64_t p0_pos, uint32_t p0_state, uint32_t p0_sym) const
    {
        SpanResult r;
        uint32_t   s = s0;
        r.sp = sp0;
        r.last_pos = r.last_sp = -1;
        r.last_state = r.last_sym = 0;
        r.pend_pos = p0_pos;
        r.pend_state = p0_state;
        r.pend_sym = p0_sym;
        for (uint32_t b = 0; b < len; b++) {
            const uint32_t sym = d->cls_map[data[gpos + b]];
            const uint32_t t2 = tr2[s * nsym + sym], next = t2 & 0xffu, kind = t2 >> 8;
            if (kind && next == 0) {
                const bool pop = kind == K_POP_FULL && gpos + b > 0;
                r.last_pos = gpos + b;
                r.last_state = s;
                r.last_sym = sym;
                r.last_sp = r.sp;
                r.sp = pop ? gpos + b : gpos + b + 1;
                r.pend_pos = -1;
                s = d->init[variant_of(data[r.sp - 1])];
                if (pop) s = tr2[s * nsym + sym] & 0xffu;
            } else if (kind) {
                r.pend_pos = gpos + b;
                r.pend_state = s;
                r.pend_sym = sym;
                s = next;
            } else if (next == 0 && gpos + b > 0) {
                r.last_pos = r.pend_pos;
                r.last_state = r.pend_state;
                r.last_sym = r.pend_sym;
                r.last_sp = r.sp;
                r.sp = gpos + b;
                s = d->init[variant_of(data[r.sp - 1])];
                const uint32_t t3 = tr2[s * nsym + sym];
                r.pend_pos = -1;
                if ((t3 >> 8) && (t3 & 0xffu) == 0) {
                    r.last_pos = gpos + b;
                    r.last_state = s;
                    r.last_sym = sym;
                    r.last_sp = r.sp;
                    r.sp = gpos + b + 1;
                    s = d->init[variant_of(data[r.sp - 1])];
                } else {
                    if (t3 >> 8) {
                        r.pend_pos = gpos + b;
                        r.pend_state = s;
                        r.pend_sym = sym;
                    }
                    s = t3 & 0xffu;
                }
            } else {
                s = next;
            }
        }
        return r;
    }
    void note_span(int64_t pos, uint32_t len, uint32_t s0, uint32_t cnt)
    {
        fl &= ~F_HAS_EV;
        if (f(F_SP_DIRTY)) {
            fcA_pos = fcB_pos;
            fcA_len = fcB_len;
            fcA_s0 = fcB_s0;
        } else {
            fc_sp0 = cur_sp;
            fcA_pos = -1;
        }
        fcB_pos = pos;
        fcB_len = len;
        fcB_s0 = s0;
        if (f(F_PEND_LAZY)) {
            fcB_pos = lz_pos;
            fcB_len = len + (f(F_LZ_GROUP) ? 16u : 64u);
            fcB_s0 = lz_s0;
        }
        fl |= F_SP_DIRTY;
        count += cnt;
    }
    void fast_span_done(int64_t pos, uint32_t len, uint32_t s0, uint32_t sum, bool fresh1)
    {
        if (!F.any_fresh) {
            if (sum) note_span(pos, len, s0, sum);
            return;
        }
        if (sum == 0) {
            fl &= ~F_PEND_LAZY;
            return;
        }
        if (sum & 127u) note_span(pos, len, s0, sum & 127u);
        else fl &= ~F_HAS_EV;
        if (fresh1) {
            fl |= F_PEND_LAZY;
            if (len == 16u) fl |= F_LZ_GROUP; else fl &= ~F_LZ_GROUP;
            lz_pos = pos;
            lz_s0 = s0;
        } else {
            fl &= ~F_PEND_LAZY;
        }
    }
    void settle()
    {
        if (!(fl & (F_SP_DIRTY | F_PEND_LAZY))) return;
        if (f(F_SP_DIRTY)) {
            int64_t sp0 = fc_sp0;
            if (fcA_pos >= 0) sp0 = resolve(fcA_pos, fcA_len, fcA_s0, -1, -1, 0, 0).sp;
            const SpanResult r = resolve(fcB_pos, fcB_len, fcB_s0, sp0, ev_pos, ev_state, ev_sym);
            if (r.last_pos >= 0) {
                fl |= F_LM_VALID;
                lm_pos = r.last_pos;
                lm_state = r.last_state;
                lm_sym = r.last_sym;
                lm_sp = r.last_sp;
                const bool anchored = r.last_sp < fcB_pos && r.last_pos - fcB_pos <= 256;
                lm_apos = anchored ? fcB_pos : -1;
                lm_astate = anchored ? fcB_s0 : 0u;
            }
            cur_sp = r.sp;
            fl &= ~F_SP_DIRTY;
        }
        if (f(F_PEND_LAZY)) {
            const SpanResult r = resolve(lz_pos, f(F_LZ_GROUP) ? 16u : 64u, lz_s0, -1, -1, 0, 0);
            fl &= ~F_PEND_LAZY;
            if (r.pend_pos >= 0) {
                fl |= F_HAS_EV;
                ev_pos = r.pend_pos;
                ev_state = r.pend_state;
                ev_sym = r.pend_sym;
                ev_kind = tr2[r.pend_state * nsym + r.pend_sym] >> 8;
                ev_sp = cur_sp;
                const bool anchored = cur_sp < lz_pos;
                ev_apos = anchored ? lz_pos : -1;
                ev_astate = anchored ? lz_s0 : 0u;
            }
        }
    }
    void run()
    {
        st = d->init[SRE_DFA_INIT_START];
        cur_sp = 0;
        const uint32_t L = span;
        for (int64_t base = 0; base < n && !f(F_FINISHED); base += (L ? L : 16)) {
            const int64_t step = L ? L : 16;
            const int64_t end = base + step <= n ? base + step : n;
            bool          exact = L == 0 || end != base + step || f(F_SKIP_NEXT);
            uint32_t      cur = st, sum = 0;
            bool          fresh1 = false;
            if (!exact) {
                for (uint32_t j = 0; j < step / F.stride; j++) {
                    uint32_t idx = 0;
                    for (uint32_t u = 0; u < F.stride; u++) {
                        const uint32_t c = data[base + j * F.stride + u];
                        idx |= (F.bits == 8 ? c : (uint32_t) d->cls_map[c]) << (u * F.bits);
                    }
                    const uint32_t e = F.fast[(size_t) cur * 256 + idx];
                    if (e & SRE_FAST_SLOW) {
                        exact = true;
                        break;
                    }
                    sum += ((e >> SRE_FAST_CNT_SHIFT) & SRE_FAST_CNT_MASK) + ((e & SRE_FAST_EVT) ? 128u : 0u);
                    fresh1 = (e & SRE_FAST_NEXT_FRESH) != 0;
                    cur = e / SRE_FAST_ROW_BYTES;
                }
            }
            anchor_pos = base;
            anchor_state = st;
            if (exact) {
                settle();
                slow_run(base, end);
            } else {
                fast_span_done(base, (uint32_t) step, st, sum, fresh1);
                st = cur;
                fast_bytes += step;
            }
        }
        if (!f(F_FINISHED)) {
            settle();
            anchor_pos = -1;
            slow_run(n, n + 1);
        }
        settle();
    }
};

}  // namespace

extern "C" {

/*
 * One COUNT lane over the whole stream.  span: 64 (round path), 16 (group path), 0 (every byte on the exact
 * path).  out[0] = count, out[1] = end of the last completed match (-1 none), out[2] = start of its search
 * (-1 unknown), out[3] = flags (8: the iteration ended with SRE_ERROR, 16: unresolved), out[4] = bytes taken
 * by fast entries, out[5] = 1 when the table has FRESH states, out[6] / out[7] = the anchor handed to the capture
 * walker with the last match (position, state; -1: none), out[8] = the position of the last match's event.
 */
void scan_sim_count(void *dv, const uint8_t *data, int64_t n, int span, int64_t *out)
{
    Lane L;
    L.d = static_cast<const sre_dfa_t *>(dv);
    sre_scan_fast_build(L.d, SRE_HIP_PIKE_COUNT, &L.F);
    L.nsym = L.d->ncls + 1;
    L.word_restart = L.d->init[SRE_DFA_INIT_RESTART_WORD] != L.d->init[SRE_DFA_INIT_RESTART];
    L.tr2.resize((size_t) L.d->nstates * L.nsym);
    for (size_t i = 0; i < L.tr2.size(); i++) {
        const sre_dfa_trans_t &a = L.d->trans[i];
        const uint32_t kind = a.ev_kind == SRE_DFA_EV_DONE ? (a.ev_empty ? K_DONE_EMPTY : K_DONE)
                            : a.ev_kind == SRE_DFA_EV_POP ? (a.ev_empty ? K_POP : K_POP_FULL) : 0;
        L.tr2[i] = (uint16_t) (a.next | (kind << 8));
    }
    L.data = data;
    L.n = n;
    L.span = (uint32_t) span;
    L.run();
    out[0] = L.count;
    out[1] = -1;
    out[2] = -1;
    if (L.f(F_LM_VALID)) {
        const uint32_t k = L.tr2[L.lm_state * L.nsym + L.lm_sym] >> 8;
        out[1] = (k == K_POP || k == K_POP_FULL) ? L.lm_pos : L.lm_pos + 1;
        out[2] = L.lm_sp;
    }
    out[3] = L.fl & (F_ERROR | F_UNRESOLVED);
    out[4] = L.fast_bytes;
    out[5] = L.F.any_fresh;
    out[6] = L.f(F_LM_VALID) ? L.lm_apos : -1;
    out[7] = L.lm_astate;
    out[8] = L.f(F_LM_VALID) ? L.lm_pos : -1;
}

/* the state of the ONE search that starts at sp, in front of position q (no restarts: the automaton alone) */
uint32_t scan_sim_state_at(void *dv, const uint8_t *data, int64_t sp, int64_t q)
{
    const sre_dfa_t *d = static_cast<const sre_dfa_t *>(dv);
    const uint32_t   word_restart = d->init[SRE_DFA_INIT_RESTART_WORD] != d->init[SRE_DFA_INIT_RESTART];
    uint32_t         v = SRE_DFA_INIT_START;
    if (sp > 0) {
        const uint32_t c = data[sp - 1];
        v = c == '\n' ? 1u : (word_restart && sre_isword(c)) ? 3u : 2u;
    }
    uint32_t st = d->init[v];
    for (int64_t p = sp; p < q && st != 0; p++) st = d->t(st, d->cls_map[data[p]]).next;
    return st;
}

}
