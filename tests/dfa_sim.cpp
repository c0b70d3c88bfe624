/*
 * tests/dfa_sim.cpp — TEST-ONLY host model of the table-driven scanner.
 *
 * Walks the step automaton (sregex_amd/csrc/sre_dfa.cpp) sequentially over a
 * buffer and reconstructs captures from the lineage tables exactly the way the
 * device kernels do (forward state trace, backward parent walk).  It lets the
 * CPU test-suite check the automaton and the capture reconstruction against
 * the oracle on every admitted reference block without a GPU.  It is
 * compiled by tests/test_dfa_model.py into tests/_build/ and is not part of,
 * nor linked into, the product library.
 */
#include "sre_dfa.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

struct Search {
    bool     poisoned;  /* the skip ran to the end of input: eof step not executed,
                           threads still listed => next exec is SRE_ERROR (:616-622) */
    int64_t  rc;        /* regex id, or -5 */
    int64_t  term;      /* position at which the search stopped */
    int64_t  ev_pos;    /* position of the final match event */
    uint32_t ev_state;  /* state BEFORE the event transition */
    uint32_t ev_sym;
};

/* one reference exec(): from `sp` (first byte processed) with initial list
 * `variant`, until the list dies or end of input; records the state before
 * every position in `trace` (trace[p - sp]). */
Search run_search(const sre_dfa_t *d, const uint8_t *data, int64_t n, int64_t sp, int variant,
                  std::vector<uint32_t> *trace)
{
    Search   r = {false, -5, sp, -1, 0, 0};
    uint32_t s = d->init[variant];
    if (trace) trace->clear();
    for (int64_t p = sp; p <= n; p++) {
        if (s == SRE_DFA_DEAD) break;
        if (p == n && r.poisoned) break;       /* "if (sp == last) break" at :304-306 */
        uint32_t sym = p < n ? d->cls_map[data[p]] : d->ncls;
        const sre_dfa_trans_t &t = d->t(s, sym);
        r.poisoned = (t.skipped != 0);
        if (trace) trace->push_back(s);
        if (t.ev_kind != SRE_DFA_EV_NONE) {
            r.rc = t.ev_regex;
            r.ev_pos = p;
            r.ev_state = s;
            r.ev_sym = sym;
        }
        s = t.next;
        r.term = p;
    }
    return r;
}

/* internal capture vector of the winning thread (nslots entries, -1 = unset) */
void captures(const sre_dfa_t *d, const uint8_t *data, int64_t sp, int variant, const Search &r,
              const std::vector<uint32_t> &trace, int64_t *vec)
{
    const uint32_t nslots = d->nslots;
    uint64_t       unresolved = nslots >= 64 ? ~0ull : ((1ull << nslots) - 1);
    for (uint32_t k = 0; k < nslots; k++) vec[k] = -1;

    const sre_dfa_trans_t &te = d->t(r.ev_state, r.ev_sym);
    uint32_t               j = te.ev_src;
    if (te.ev_kind == SRE_DFA_EV_DONE) {
        for (uint32_t k = 0; k < nslots; k++) {
            if ((te.ev_saves >> k) & 1) {
                vec[k] = r.ev_pos + 1;
                unresolved &= ~(1ull << k);
            }
        }
    }
    /* SAVEs of a look-ahead splice in front of the event: the position itself */
    for (uint32_t k = 0; k < nslots; k++) {
        if (((te.ev_early & unresolved) >> k) & 1) {
            vec[k] = r.ev_pos;
            unresolved &= ~(1ull << k);
        }
    }
    /* thread j lives in the list of state trace[p - sp] at position p */
    for (int64_t p = r.ev_pos; unresolved; p--) {
        uint32_t s_here = trace[(size_t) (p - sp)];
        if (d->list_pcs[d->list_off[s_here] + j] == 1) break;   /* the ".*?" ANY: nothing saved yet */
        const sre_dfa_trans_t *t;
        int64_t                val;
        if (p == sp) {
            t = &d->trans[(size_t) d->nstates * (d->ncls + 1) + variant];   /* initial closure */
            val = sp;
        } else {
            uint32_t s_prev = trace[(size_t) (p - 1 - sp)];
            t = &d->t(s_prev, d->cls_map[data[p - 1]]);
            val = p;
        }
        uint64_t m = d->lin_saves[t->lin_off + j] & unresolved;
        for (uint32_t k = 0; k < nslots; k++) {
            if ((m >> k) & 1) vec[k] = val;
        }
        unresolved &= ~m;
        m = d->lin_early[t->lin_off + j] & unresolved;      /* saved by a splice before the byte */
        for (uint32_t k = 0; k < nslots; k++) {
            if ((m >> k) & 1) vec[k] = val - 1;
        }
        unresolved &= ~m;
        if (p == sp) break;
        j = d->lin_parent[t->lin_off + j];
        if (j == SRE_DFA_NO_PARENT) break;       /* re-seeded by the leading-byte skip */
    }
}

}  // namespace

extern "C" {

void *dfa_sim_build(const sre_program_t *prog, uint32_t max_states, const char **why)
{
    return sre_dfa_build(prog, max_states, why);
}

void dfa_sim_free(void *d) { sre_dfa_free(static_cast<sre_dfa_t *>(d)); }

uint32_t dfa_sim_nstates(void *d) { return static_cast<sre_dfa_t *>(d)->nstates; }
uint32_t dfa_sim_ncls(void *d) { return static_cast<sre_dfa_t *>(d)->ncls; }
uint32_t dfa_sim_max_threads(void *d) { return static_cast<sre_dfa_t *>(d)->max_threads; }
int dfa_sim_has_lookahead(void *d) { return static_cast<sre_dfa_t *>(d)->has_lookahead; }

/*
 * The find-all iteration (or a single exec when max_matches == 1) on ONE
 * logical context.  Writes per match: regex id + ovector[nov] (caller slices of
 * the internal vector, as sre_vm_pike.c:945-989).  Returns the match count.
 */
int64_t dfa_sim_findall(void *dv, const sre_program_t *prog, const uint8_t *data, int64_t n,
                        int64_t *out, int64_t nov, int64_t max_matches)
{
    const sre_dfa_t      *d = static_cast<sre_dfa_t *>(dv);
    std::vector<uint32_t> trace;
    std::vector<int64_t>  vec(d->nslots + 1);
    int64_t               count = 0, chunk = 0;
    bool                  empty_capture = false, seen_newline = false, seen_word = false, ctx_eof = false;
    auto isword = [](uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; };
    /* a re-armed search: ^ from seen_newline, \b / \B from seen_word (sre_vm_pike.c:472-473, 586-601) */
    auto rearmed = [&](bool nl, bool word) {
        return nl ? SRE_DFA_INIT_RESTART_NL : word ? SRE_DFA_INIT_RESTART_WORD : SRE_DFA_INIT_RESTART;
    };

    while (count < max_matches) {
        if (ctx_eof) return -(count + 1);      /* SRE_ERROR ends the iteration */
        int64_t sp = chunk;
        int     variant;
        if (empty_capture) {                                   /* sre_vm_pike.c:179-196 */
            if (chunk == n) break;
            sp = chunk + 1;
            variant = rearmed(data[chunk] == '\n', isword(data[chunk]));
        } else if (chunk == 0) {
            variant = SRE_DFA_INIT_START;
        } else {
            variant = rearmed(seen_newline, seen_word);
        }
        Search r = run_search(d, data, n, sp, variant, &trace);
        if (r.rc < 0) break;
        captures(d, data, sp, variant, r, trace, vec.data());

        uint64_t ofs = 0;
        for (int64_t i = 0; i < r.rc; i++) ofs += prog->multi_ncaps[i] + 1;
        ofs *= 2;
        int64_t  ncopy = 2 * ((int64_t) prog->multi_ncaps[r.rc] + 1);
        int64_t *rec = out + count * (nov + 1);
        rec[0] = r.rc;
        for (int64_t k = 0; k < nov; k++) rec[1 + k] = k < ncopy ? vec[ofs + k] : -1;

        /* :586-601 — flags for the next search come from slot 1 of the match */
        if (vec[1] >= 0 && vec[1] > chunk) {
            seen_newline = data[vec[1] - 1] == '\n';
            seen_word = isword(data[vec[1] - 1]);
        }
        int64_t start = vec[ofs], end = vec[ofs + 1];
        empty_capture = (start == end);
        /* what the COUNT scan goes by instead of captures: the event transition's own flag */
        if ((d->t(r.ev_state, r.ev_sym).ev_empty != 0) != empty_capture) return -1000000;
        chunk = end;
        count++;
        ctx_eof = r.poisoned;
    }
    return count;
}

/* match / no match (sre_vm_thompson_exec with eof): any match event at all */
int64_t dfa_sim_thompson(void *dv, const uint8_t *data, int64_t n)
{
    const sre_dfa_t *d = static_cast<sre_dfa_t *>(dv);
    Search           r = run_search(d, data, n, 0, SRE_DFA_INIT_START, NULL);
    return r.rc >= 0 ? 0 : -5;
}

}  // extern "C"
