/*
 * sre_scan_fast.cpp — the scanner's FAST TABLE (sre_scan_fast.h): the step automaton (sre_dfa.cpp) flattened into
 * one 32-bit entry per (state, 8-bit index of packed byte classes), with the find-all iteration's restarts folded
 * in for COUNT scans.  Host code without a device call: sre_scan_host.cpp uploads what this builds, and
 * tests/scan_sim.cpp walks it on the CPU against the oracle.
 */
#include "sre_scan_fast.h"
#include <sregex_hip.h>
#include <stdlib.h>
#include <string.h>

void
sre_scan_fast_build(const sre_dfa_t *d, int mode, sre_scan_fast_t *out)
{
    out->any_fresh = 0;
    /* fast table: `stride` bytes per lookup through packed byte classes */
    uint32_t bits = 8;
    if (d->ncls <= 2) bits = 1;
    else if (d->ncls <= 4) bits = 2;
    else if (d->ncls <= 16) bits = 4;
    const uint32_t stride = 8 / bits;
    /* is class k the newline?  (classes separate it whenever ^ is in the program) */
    std::vector<uint8_t> rep_is_nl(d->ncls + 1, 0);
    if (d->has_caret) rep_is_nl[d->cls_map[(unsigned char) '\n']] = 1;
    /* ... a word byte?  (classes separate them whenever \b or \B is) */
    std::vector<uint8_t> rep_is_word(d->ncls + 1, 0);
    if (d->init[SRE_DFA_INIT_RESTART_WORD] != d->init[SRE_DFA_INIT_RESTART]) {
        for (unsigned c = 0; c < 256; c++) {
            if (sre_isword(c)) rep_is_word[d->cls_map[c]] = 1;
        }
    }
    /* COUNT: a match that a look-ahead assertion completes (`foo$`, `\bfoo\b`) ends IN FRONT of the byte that
     * decided it; when nothing outlives it the next search starts AT that byte, from the initial list the byte
     * in front of it selects (sre_vm_pike.c:586-601, :624-628).  Inside an index that byte is the previous
     * sub-step's; at its first sub-step it is whatever byte led into the row's state — known when every way
     * into the state (transitions, its role as an initial list) agrees on the initial list it selects. */
    auto restart_of = [&](uint32_t k) {
        return d->init[rep_is_nl[k] ? SRE_DFA_INIT_RESTART_NL : rep_is_word[k] ? SRE_DFA_INIT_RESTART_WORD : SRE_DFA_INIT_RESTART];
    };
    const uint32_t RESTART_NONE = 0xfffffffeu, RESTART_MIXED = 0xffffffffu;
    std::vector<uint32_t> restart_in(d->nstates, RESTART_NONE);
    {
        auto merge = [&](uint32_t s, uint32_t r) {
            if (s == SRE_DFA_DEAD || s >= d->nstates) return;
            restart_in[s] = restart_in[s] == RESTART_NONE || restart_in[s] == r ? r : RESTART_MIXED;
        };
        for (uint32_t s = 1; s < d->nstates; s++) {
            for (uint32_t k = 0; k < d->ncls; k++) merge(d->t(s, k).next, restart_of(k));
        }
        merge(d->init[SRE_DFA_INIT_RESTART], d->init[SRE_DFA_INIT_RESTART]);
        merge(d->init[SRE_DFA_INIT_RESTART_NL], d->init[SRE_DFA_INIT_RESTART_NL]);
        merge(d->init[SRE_DFA_INIT_RESTART_WORD], d->init[SRE_DFA_INIT_RESTART_WORD]);
    }
    const bool fold_pop = getenv("SRE_HIP_NO_POP_FOLD") == NULL;     /* (experiment knob) */
    /* COUNT: a pending match that GROWS (`[a-z]+` inside a word: every byte completes a longer match, the
     * list lives on) used to send every such byte through the kernel's exact path — find-all of any pattern
     * with a greedy tail ran at 0.014 of peak (tools/floor_probe.py).  A state is FRESH when every way into
     * it is a transition that records a non-empty match ending with the byte just consumed: the pending
     * match of a lane in a fresh state is then known without having been recorded (its end is the previous
     * byte; state and symbol of its event come from replaying the last round, sre_hip_scan.hip settle()).
     * Such transitions stay in the fast table (SRE_FAST_EVT), and so does the list dying in a fresh state
     * without a new event: the pending match completes, it ends right here, and the next search reads the
     * byte again (as above).  A step out of a fresh state that neither records a match nor kills the list
     * would leave the pending match behind unrecorded: those take the exact path. */
    const bool fold_grow = getenv("SRE_HIP_NO_GROW_FOLD") == NULL;   /* (experiment knob) */
    std::vector<uint8_t> fresh(d->nstates, 0);
    if (mode == SRE_HIP_PIKE_COUNT && fold_grow) {
        std::vector<uint8_t> any_in(d->nstates, 0), bad_in(d->nstates, 0);
        for (uint32_t s = 1; s < d->nstates; s++) {
            for (uint32_t k = 0; k <= d->ncls; k++) {
                const sre_dfa_trans_t &tr = d->t(s, k);
                if (tr.next == SRE_DFA_DEAD || tr.next >= d->nstates) continue;
                any_in[tr.next] = 1;
                if (k == d->ncls || tr.ev_kind != SRE_DFA_EV_DONE || tr.ev_empty || tr.skipped) bad_in[tr.next] = 1;
            }
        }
        for (int v = 0; v < SRE_DFA_NINIT; v++) bad_in[d->init[v]] = 1;
        for (uint32_t s = 1; s < d->nstates; s++) {
            fresh[s] = any_in[s] && !bad_in[s] && d->matched[s];
            if (fresh[s]) out->any_fresh = 1;
        }
    }
    auto build_fast = [&](int fmode) {
    std::vector<uint32_t> fast((size_t) d->nstates * 256);
    for (uint32_t s = 0; s < d->nstates; s++) {
        for (unsigned idx = 0; idx < 256; idx++) {
            uint32_t st = s, flags = 0, cnt = 0;
            for (uint32_t sub = 0; sub < stride && !(flags & SRE_FAST_SLOW); sub++) {
                /* sub-step `sub` consumes input byte `sub` of the group */
                const uint32_t k = bits == 8 ? d->cls_map[idx] : ((idx >> (sub * bits)) & ((1u << bits) - 1));
                if (st == SRE_DFA_DEAD || k >= d->ncls) {
                    flags |= SRE_FAST_SLOW;
                    break;
                }
                const sre_dfa_trans_t &tr = d->t(st, k);
                if (fmode == SRE_HIP_PIKE_COUNT && tr.ev_kind == SRE_DFA_EV_DONE
                    && tr.next == SRE_DFA_DEAD && !tr.ev_empty)
                {
                    /* a non-empty match completes and nothing outlives it: the
                     * next search starts at the next byte (sre_vm_pike.c:624-628) */
                    /* the byte in front of that search is the one just consumed */
                    st = restart_of(k);
                    cnt++;
                } else if (fold_pop && fmode == SRE_HIP_PIKE_COUNT && tr.ev_kind == SRE_DFA_EV_POP
                           && tr.next == SRE_DFA_DEAD && tr.ev_empty)
                {
                    /* an EMPTY match in front of this byte (`\b`, `$`, `x*` where no x is): the caller skips the
                     * byte (sre_vm_pike.c:179-196) and the next search starts behind it */
                    st = restart_of(k);
                    cnt++;
                } else if (fold_pop && fmode == SRE_HIP_PIKE_COUNT && tr.ev_kind == SRE_DFA_EV_POP
                           && tr.next == SRE_DFA_DEAD && !tr.ev_empty)
                {
                    /* ... or in front of this byte, which the next search reads again */
                    uint32_t r = RESTART_MIXED;
                    if (sub > 0) {
                        r = restart_of((idx >> ((sub - 1) * bits)) & ((1u << bits) - 1));
                    } else if (restart_in[s] != RESTART_NONE) {
                        r = restart_in[s];
                    }
                    bool ok = r != RESTART_MIXED && r != SRE_DFA_DEAD;
                    if (ok) {
                        const sre_dfa_trans_t &tr2 = d->t(r, k);
                        ok = tr2.ev_kind == SRE_DFA_EV_NONE && tr2.next != SRE_DFA_DEAD;
                        if (ok) {
                            st = tr2.next;
                            cnt++;
                        }
                    }
                    if (!ok) flags |= SRE_FAST_SLOW;
                } else if (fmode == SRE_HIP_PIKE_COUNT && tr.ev_kind == SRE_DFA_EV_DONE && !tr.ev_empty && !tr.skipped
                           && tr.next != SRE_DFA_DEAD && fresh[tr.next])
                {
                    /* the pending match grows */
                    st = tr.next;
                    flags |= SRE_FAST_EVT;
                } else if (fmode == SRE_HIP_PIKE_COUNT && tr.ev_kind == SRE_DFA_EV_NONE && tr.next == SRE_DFA_DEAD && fresh[st]) {
                    /* the list dies, the pending match ends right here: the next search reads this byte again */
                    uint32_t r = RESTART_MIXED;
                    if (sub > 0) {
                        r = restart_of((idx >> ((sub - 1) * bits)) & ((1u << bits) - 1));
                    } else if (restart_in[s] != RESTART_NONE) {
                        r = restart_in[s];
                    }
                    bool ok = r != RESTART_MIXED && r != SRE_DFA_DEAD;
                    if (ok) {
                        const sre_dfa_trans_t &tr2 = d->t(r, k);
                        const bool             plain = tr2.ev_kind == SRE_DFA_EV_NONE && tr2.next != SRE_DFA_DEAD;
                        const bool             grows = tr2.ev_kind == SRE_DFA_EV_DONE && !tr2.ev_empty && !tr2.skipped
                                                       && tr2.next != SRE_DFA_DEAD && fresh[tr2.next];
                        /* ... or completes a match of its own at once: this one byte, or an empty one in front of it */
                        const bool             again = tr2.next == SRE_DFA_DEAD
                                                       && ((tr2.ev_kind == SRE_DFA_EV_DONE && !tr2.ev_empty)
                                                           || (fold_pop && tr2.ev_kind == SRE_DFA_EV_POP && tr2.ev_empty));
                        ok = plain || grows || again;
                        if (ok) {
                            st = again ? restart_of(k) : tr2.next;
                            cnt += again ? 2 : 1;
                            if (grows) flags |= SRE_FAST_EVT;
                        }
                    }
                    if (!ok) flags |= SRE_FAST_SLOW;
                } else if (tr.ev_kind != SRE_DFA_EV_NONE || tr.next == SRE_DFA_DEAD
                           || (fmode == SRE_HIP_PIKE_COUNT && fresh[st]))
                {
                    flags |= SRE_FAST_SLOW;
                } else {
                    st = tr.next;
                }
            }
            if (flags & SRE_FAST_SLOW) {
                st = s;
                cnt = 0;
                flags = SRE_FAST_SLOW;
            } else if (fmode == SRE_HIP_PIKE_COUNT && st < d->nstates && fresh[st]) {
                flags |= SRE_FAST_NEXT_FRESH;
            }
            fast[(size_t) s * 256 + idx] = st * SRE_FAST_ROW_BYTES | flags
                                           | (cnt << SRE_FAST_CNT_SHIFT);
        }
    }
    return fast;
    };
    out->fast = build_fast(mode);
    out->fast_plain = mode == SRE_HIP_PIKE_COUNT ? build_fast(SRE_HIP_PIKE_FIRST) : out->fast;
    out->bits = bits;
    out->stride = stride;
    out->fresh = fresh;
}
