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
#include <string.h>
#include <stdint.h>

extern "C" {

void *nfa_sim_build(const sre_program_t *prog, const char **why) { return sre_nfa_build2(prog, SRE_NFA_SA_OFF, why); }
void *nfa_sim_build2(const sre_program_t *prog, unsigned sa_options, const char **why) { return sre_nfa_build2(prog, sa_options, why); }
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


/* ---- the shift-and form (sre_nfa.h): one step as the device kernel takes it */

static inline uint64_t
sa_step(const sre_nfa_sa_t *a, uint64_t S, unsigned byte, uint64_t *t_out)
{
    const uint64_t t = S & a->accept[byte];
    const uint64_t ts = a->masked ? t & a->shift_src : t;
    uint64_t       sh;
    if (!a->w64) sh = (uint64_t) (uint32_t) ((uint32_t) ts << 1);
    else if (a->carry) sh = ts << 1;
    else sh = ((uint64_t) (uint32_t) ((uint32_t) (ts >> 32) << 1) << 32) | (uint32_t) ((uint32_t) ts << 1);
    uint64_t r = sh | (t & a->self) | a->seed;
    for (uint32_t k = 0; k < a->nlut; k++) r |= a->lut[(size_t) k * 256 + ((t >> (8 * a->hot[k])) & 0xff)];
    *t_out = t;
    return r;
}

/* info[0..9] = has form, nbits, w64, carry, masked, evacc, nlut, cost, threads (valid bits), look-ahead assertions */
void nfa_sim_sa_info(void *h, int32_t *info)
{
    const sre_nfa_t *n = static_cast<sre_nfa_t *>(h);
    for (int i = 0; i < 10; i++) info[i] = 0;
    if (!n->sa) return;
    const sre_nfa_sa_t *a = n->sa;
    info[0] = 1; info[1] = (int32_t) a->nbits; info[2] = (int32_t) a->w64; info[3] = (int32_t) a->carry;
    info[4] = (int32_t) a->masked; info[5] = (int32_t) a->evacc; info[6] = (int32_t) a->nlut; info[7] = (int32_t) a->cost;
    info[8] = __builtin_popcountll(a->valid);
    info[9] = (int32_t) a->nassert;
}

/* Runs the plain form and the shift-and form side by side over the buffer.  out[0] / out[1] as
 * nfa_sim_run (from the shift-and form), out[2] = first position at which the two disagree (-1:
 * never): the state sets (mapped through bit_of), the "only the .*? thread consumed" test, the event. */
void nfa_sim_run_sa(void *h, const uint8_t *data, int64_t n, int variant, int64_t *out)
{
    const sre_nfa_t    *g = static_cast<sre_nfa_t *>(h);
    const sre_nfa_sa_t *a = g->sa;
    out[0] = out[1] = out[2] = -1;
    if (!a) return;
    uint64_t S = a->init[variant], G = g->init[variant];
    int64_t  clean = 0, ev = -1, bad = -1;
    uint32_t prev = SRE_NFA_KIND_EDGE;
    auto map = [&](uint64_t gm) {
        uint64_t m = 0;
        for (uint32_t i = 0; i < g->nbits; i++) {
            if (((gm >> i) & 1) && a->bit_of[i] >= 0) m |= 1ull << a->bit_of[i];
        }
        return m;
    };
    if (map(G) != (S & a->valid)) bad = 0;
    for (int64_t p = 0; p <= n; p++) {
        const uint32_t cur = p < n ? (g->kind[data[p]] & 3u) : (uint32_t) SRE_NFA_KIND_EDGE;
        if (a->nassert) {
            /* look-ahead assertions that hold list their continuation at this position */
            S |= a->expand[(size_t) (prev * 4 + cur) * 256 + ((S >> (8 * a->assert_byte)) & 0xff)];
            G |= g->expand[(size_t) (prev * 4 + cur) * 256 + ((G >> (8 * g->assert_slice)) & 0xff)];
            const bool e1 = (S & a->match_bits) != 0, e2 = (G & g->match_bits) != 0;
            if (bad < 0 && (e1 != e2 || (!e1 && map(G) != (S & a->valid & ~a->match_bits)))) bad = p + 1;
            if (e1) {
                ev = p;
                break;
            }
        }
        if (p == n) break;
        uint64_t t, gt = G & g->accept[data[p]], gr = 0;
        S = sa_step(a, S, data[p], &t);
        for (uint32_t k = 0; k < g->nslices; k++) gr |= g->follow[(size_t) k * 256 + ((gt >> (8 * k)) & 0xff)];
        G = gr;
        const bool event = a->evacc ? (t & a->msrc) != 0 : (S & a->match_bits) != 0;
        const bool gevent = (G & g->match_bits) != 0;
        const bool cl = (t & ~a->any_bits) == 0, gcl = (gt & ~g->any_bits) == 0;
        if (bad < 0 && (event != gevent || (!event && (cl != gcl || map(G) != (S & a->valid & ~a->match_bits))))) bad = p + 1;
        if (event) {
            ev = p;
            break;
        }
        if (cl) clean = p + 1;
        prev = cur;
    }
    out[0] = ev;
    out[1] = clean;
    out[2] = bad;
}


/* The thread set after stepping S over data[0, n) with no regard to events (what a segment does to an entry set:
 * sre_hip_nfa.hip sre_k_nfa_seg_matrix), by the plain form (sa == 0) or the shift-and form; prev0 = the kind of
 * the byte in front (SRE_NFA_KIND_*).  The exact-entry fallback of the NFA tier rests on this being a
 * union-homomorphism of S. */
uint64_t nfa_sim_walk_set(void *h, int sa, uint64_t S, const uint8_t *data, int64_t n, uint32_t prev0)
{
    const sre_nfa_t    *g = static_cast<sre_nfa_t *>(h);
    const sre_nfa_sa_t *a = g->sa;
    uint32_t            prev = prev0;
    for (int64_t p = 0; p < n; p++) {
        const uint32_t cur = g->kind[data[p]] & 3u;
        if (sa) {
            if (a->nassert) S |= a->expand[(size_t) (prev * 4 + cur) * 256 + ((S >> (8 * a->assert_byte)) & 0xff)];
            uint64_t t;
            S = sa_step(a, S, data[p], &t);
        } else {
            if (g->nassert) S |= g->expand[(size_t) (prev * 4 + cur) * 256 + ((S >> (8 * g->assert_slice)) & 0xff)];
            const uint64_t t = S & (g->accept[data[p]] | g->match_bits);
            uint64_t       r = t & g->match_bits;
            for (uint32_t k = 0; k < g->nslices; k++) r |= g->follow[(size_t) k * 256 + ((t >> (8 * k)) & 0xff)];
            S = r;
        }
        prev = cur;
    }
    return S;
}
uint64_t nfa_sim_valid_bits(void *h, int sa)
{
    const sre_nfa_t *g = static_cast<sre_nfa_t *>(h);
    if (sa) return g->sa ? g->sa->valid : 0;
    return g->nbits >= 64 ? ~0ull : ((1ull << g->nbits) - 1);
}


/* One exec() of the Thompson WAVE kernel (sre_hip_vm.hip thompson_wave_run: lanes = threads, the live set a 64-bit
 * mask, S' = ballot(pred[lane] & S & accept[byte]), 64 bytes a block) on the mask *S_io, with its STABLE RUNS: a
 * byte that maps S to itself joins a 256-bit set kept for that S, and from the second such step in a row the bytes
 * of the set that follow are skipped — inside the block by one ballot, across blocks 512 bytes at a time.
 * ff == 0: without the skipping.  Returns 0 (match), -5 (declined), -2 (again); *skipped += bytes not stepped. */
int64_t nfa_sim_thompson_wave(void *h, uint64_t *S_io, const uint8_t *input, int64_t size, int eof, int ff, int64_t *skipped)
{
    const sre_nfa_t *g = static_cast<sre_nfa_t *>(h);
    uint64_t pred[64];
    memset(pred, 0, sizeof(pred));
    for (uint32_t i = 0; i < g->nbits; i++) {
        const uint64_t f = g->follow[(size_t) (i / 8) * 256 + (1u << (i % 8))];
        for (uint32_t q = 0; q < g->nbits; q++) {
            if ((f >> q) & 1) pred[q] |= 1ull << i;
        }
    }
    const uint64_t match = g->match_bits;
    uint64_t       S = *S_io, stabS = 0;
    uint32_t       stab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool           stab_valid = false, streak = false;
    auto member = [&](uint32_t c) { return ((stab[c >> 5] >> (c & 31)) & 1u) != 0; };
    int64_t base = 0, rc = -100;
    while (base < size && rc == -100) {
        if (ff && streak && stab_valid && stabS == S) {
            bool found = false;
            while (!found && base < size) {
                for (int k = 0; k < 8 && !found; k++) {
                    for (int lane = 0; lane < 64 && !found; lane++) {
                        const int64_t p = base + k * 64 + lane;
                        if (p >= size || !member(input[p])) {
                            *skipped += k * 64 + lane;
                            base += k * 64 + lane;
                            found = true;
                        }
                    }
                }
                if (!found) {
                    *skipped += 512;
                    base += 512;
                }
            }
            if (base >= size) break;
        }
        const uint32_t nb = size - base < 64 ? (uint32_t) (size - base) : 64u;
        uint32_t       i = 0;
        while (i < nb) {
            if (S == 0) {
                rc = eof ? -5 : -2;
                break;
            }
            if (S & match) {
                rc = 0;
                break;
            }
            const uint64_t T = S & g->accept[input[base + i]];
            uint64_t       S1 = 0;
            for (uint32_t q = 0; q < 64; q++) {
                if (pred[q] & T) S1 |= 1ull << q;
            }
            i++;
            if (S1 != S) {
                S = S1;
                streak = false;
                continue;
            }
            if (streak && ff) {
                const uint32_t b = input[base + i - 1];
                if (!stab_valid || stabS != S) {
                    memset(stab, 0, sizeof(stab));
                    stabS = S;
                    stab_valid = true;
                }
                stab[b >> 5] |= 1u << (b & 31);
                if (i < nb) {
                    uint32_t run = 0;
                    while (i + run < nb && member(input[base + i + run])) run++;
                    *skipped += run;
                    i += run;
                }
            }
            streak = true;
        }
        base += nb;
    }
    *S_io = S;
    if (rc != -100) return rc;
    if (eof && (S & match)) return 0;
    return eof ? -5 : -2;
}
uint64_t nfa_sim_init0(void *h) { return static_cast<sre_nfa_t *>(h)->init[0]; }
int nfa_sim_nassert(void *h) { return (int) static_cast<sre_nfa_t *>(h)->nassert; }

}
