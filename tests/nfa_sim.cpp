/*
 * tests/nfa_sim.cpp — TEST-ONLY host model of the bit-parallel NFA scanner.
 *
 * Walks the tables of sregex_amd/csrc/sre_nfa.cpp sequentially over a buffer the
 * way the device kernel does per lane (T = S & accept[b]; S = OR of the follow
 * slices) and reports the first MATCH event and the last clean position in front
 * of it, so the CPU suite can check the set semantics against the reference
 * goldens and the oracle without a GPU.  Compiled by tests/test_nfa_model.py into
 * tests/_build/; not part of, nor linked into, the product library.
 */
#include "sre_nfa.h"
#include <stdint.h>

extern "C" {

void *nfa_sim_build(const sre_program_t *prog, const char **why) { return sre_nfa_build(prog, why); }
void nfa_sim_free(void *h) { sre_nfa_free(static_cast<sre_nfa_t *>(h)); }
uint32_t nfa_sim_nbits(void *h) { return static_cast<sre_nfa_t *>(h)->nbits; }

/* out[0] = first event step (-1 none), out[1] = last clean position <= it (every
 * position checked, not sampled), out[2] = population count high-water mark */
void nfa_sim_run(void *h, const uint8_t *data, int64_t n, int variant, int64_t *out)
{
    const sre_nfa_t *a = static_cast<sre_nfa_t *>(h);
    uint64_t S = a->init[variant];
    int64_t  clean = 0, ev = -1, hw = 0;
    uint32_t prev = SRE_NFA_KIND_EDGE;          /* start of the stream */
    for (int64_t p = 0; p <= n; p++) {
        const uint32_t cur = p < n ? (a->kind[data[p]] & 3u) : (uint32_t) SRE_NFA_KIND_EDGE;
        if (a->nassert) {
            /* look-ahead assertions that hold list their continuation at this position */
            S |= a->expand[(size_t) (prev * 4 + cur) * 256 + ((S >> (8 * a->assert_slice)) & 0xff)];
            if (S & a->match_bits) {
                ev = p;                         /* a listed MATCH is popped at this step */
                break;
            }
        }
        if (p == n) break;
        const uint64_t t = S & a->accept[data[p]];
        uint64_t       r = 0;
        for (uint32_t k = 0; k < a->nslices; k++) r |= a->follow[(size_t) k * 256 + ((t >> (8 * k)) & 0xff)];
        S = r;
        if (__builtin_popcountll(S) > hw) hw = __builtin_popcountll(S);
        if (S & a->match_bits) {
            ev = p;
            break;
        }
        if ((t & ~a->any_bits) == 0) clean = p + 1;
        prev = cur;
    }
    out[0] = ev;
    out[1] = clean;
    out[2] = hw;
}

}
