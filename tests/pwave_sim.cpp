/*
 * tests/pwave_sim.cpp — TEST-ONLY host model of the wavefront Pike step (sregex_amd/csrc/sre_hip_pwave.hip).
 *
 * Takes the tables of sre_pwave_build() (static closure lists per instruction and context) and runs
 * one whole-buffer exec of a fresh context the way the kernel does — accept test per listed thread,
 * first listed MATCH cuts the list, consuming threads in priority order load their closure list, a
 * stamp per thread makes the first arrival win, survivors are ranked in source order then closure
 * order, capture columns are handed over with the saved slots replaced — only with loops where the
 * kernel has lanes.  The CPU suite checks it against the oracle (tests/test_pwave_model.py), which
 * pins the ALGORITHM (that static lists + first-arrival-wins reproduce the reference's generation
 * tags, its SPLIT re-descent, SRE_DONE and the leading-byte skip) without a GPU.  Compiled into
 * tests/_build/; not part of, nor linked into, the product library.
 */
#include "sre_pwave.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

int     g_ff_enabled = 1;
int64_t g_ff_bytes = 0;

struct Sim {
    const sre_pwave_hdr_t   *W;
    const sre_pwave_list_t  *lists;
    const sre_pwave_entry_t *ents;
    const uint8_t           *in;
    uint32_t                 nslots, n = 0, cur = 0, stamp_cur = 0;
    int64_t                  processed = 0;
    uint32_t                 seen_newline = 0, sss = 0, initial_count = 0, has_matched = 0, poisoned = 0;
    uint32_t                 empty_capture = 0, ctx_eof = 0, first_buf = 1;
    int64_t                  matched_id = -1;
    std::vector<int64_t>     caps[2], matched;
    uint16_t                 tidv[2][64], initial[64];
    uint32_t                 stamp[64];
    uint32_t                 lab[2][64];     /* equal labels => equal capture columns */
    /* stable runs (sre_hip_pwave.hip "stable runs"): bytes whose step left list, columns and sss as they were */
    uint32_t                 stab[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stab_on = 0, stab_x = 0, step_stable = 0;

    int64_t skip_stable(int64_t pos, int64_t last)
    {
        while (pos < last && ((stab[in[pos] >> 5] >> (in[pos] & 31)) & 1)) pos++;
        return pos;
    }
    void learn(bool plain, uint32_t c, uint32_t x_in)
    {
        if (plain && step_stable && sss == x_in && g_ff_enabled) {
            if (!stab_on || stab_x != x_in) {
                memset(stab, 0, sizeof(stab));
                stab_x = x_in;
                stab_on = 1;
            }
            stab[c >> 5] |= 1u << (c & 31);
        } else {
            stab_on = 0;
        }
    }

    uint32_t ctx_at(int64_t pos) const
    {
        if (pos == 0) return processed == 0 ? 2u : (seen_newline ? 1u : 0u);
        return in[pos - 1] == '\n' ? 1u : 0u;
    }
    void seed(int64_t pos)
    {
        const sre_pwave_list_t L = lists[ctx_at(pos)];
        for (uint32_t k = 0; k < L.len; k++) {
            const sre_pwave_entry_t e = ents[L.off + k];
            tidv[cur][k] = e.tid;
            lab[cur][k] = k;
            for (uint32_t s = 0; s < nslots; s++) caps[cur][s * 64 + k] = ((e.saves >> s) & 1) ? processed + pos : -1;
        }
        n = L.len;
        if (L.sss) sss = 1;
    }
    int64_t find_first_byte(int64_t pos, int64_t last) const
    {
        for (; pos < last; pos++) {
            const uint32_t c = in[pos];
            if ((W->lead[c >> 5] >> (c & 31)) & 1) return pos;
        }
        return last;
    }
    int64_t exec(int64_t size, int64_t start, bool skip_target, int64_t *ov, uint32_t ovec_slots)
    {
        const int64_t last = size;
        int64_t       sp = 0, last_matched_pos = -1;
        bool          no_check_once = false, skip_ran_out = false;
        if (ctx_eof) return -1;
        if (empty_capture) {
            empty_capture = 0;
            if (size == 0) {
                ctx_eof = 1;
                return -5;
            }
            sp = 1;
        }
        cur = 0;
        n = 0;
        sss = 0;
        has_matched = 0;
        poisoned = 0;
        stamp_cur = 0;
        memset(stamp, 0, sizeof(stamp));
        seed(sp);
        initial_count = n;
        for (uint32_t k = 0; k + 1 < n; k++) initial[k] = tidv[cur][k];
        if (start > sp) {
            sp = start;
            if (W->nleading && skip_target) {
                sp = find_first_byte(sp, last);
                no_check_once = true;
                if (sp == last) skip_ran_out = true;
            }
            seed(sp);
        }
        stab_on = 0;
        for (; !skip_ran_out && sp <= last; sp++) {
            if (n == 0) break;
            bool     plain = true;
            uint32_t x_in = sss;
            if (no_check_once) {
                no_check_once = false;
                plain = false;
            } else {
                if (stab_on && sss == stab_x && sp < last) {
                    const int64_t p = skip_stable(sp, last);
                    g_ff_bytes += p - sp;
                    sp = p;
                }
                x_in = sss;
                if (W->nleading && sss) {
                    sss = 0;
                    bool same = (sp != last) && (n == initial_count);
                    for (uint32_t k = 0; same && k + 1 < n; k++) same = tidv[cur][k] == initial[k];
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
            const uint32_t c = sp < last ? in[sp] : 0u;
            if (step(sp, last)) last_matched_pos = matched[1];
            learn(plain && sp < last, c, x_in);
            if (sp == last) break;
        }
        if (last_matched_pos >= 0) {
            const int64_t p = last_matched_pos - processed;
            if (p > 0) seen_newline = in[p - 1] == '\n';
        }
        if (has_matched) {
            if (matched_id >= (int64_t) W->nregexes) return -1;
            const uint32_t *ncaps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(W) + W->multi_ncaps_off);
            uint32_t        ofs = 0;
            for (int64_t r = 0; r < matched_id; r++) ofs += ncaps[r] + 1;
            ofs *= 2;
            const uint32_t cnt = 2u * (ncaps[matched_id] + 1);
            for (uint32_t k = 0; k < ovec_slots; k++) ov[k] = k < cnt ? matched[ofs + k] : -1;
            if (n > 0) {
                poisoned = 1;
                ctx_eof = 1;
            }
            processed = matched[ofs + 1];
            empty_capture = matched[ofs] == matched[ofs + 1];
            return matched_id;
        }
        ctx_eof = 1;
        return -5;
    }

    /* the same context fed in CHUNKS (sre_vm_pike.c:148-689 with eof == 0: SRE_AGAIN, the temporary
     * match range and the pending match, :640-735) — what sre_k_pike_exec's wave path does */
    int64_t exec_chunk(int64_t size, bool eof, int64_t *ov, uint32_t ovec_slots, int *has_pending, int64_t *pending)
    {
        const int64_t last = size;
        int64_t       sp = 0, last_matched_pos = -1;
        *has_pending = 0;
        if (ctx_eof) return -1;
        if (empty_capture) {
            empty_capture = 0;
            if (size == 0) {
                if (eof) {
                    ctx_eof = 1;
                    return -5;
                }
                return -2;
            }
            sp = 1;
        }
        if (first_buf) {
            first_buf = 0;
            memset(stamp, 0, sizeof(stamp));
            stamp_cur = 0;
            sss = 0;
            seed(sp);
            initial_count = n;
            for (uint32_t k = 0; k + 1 < n; k++) initial[k] = tidv[cur][k];
        }
        stab_on = 0;
        for (; sp < last || (eof && sp == last); sp++) {
            if (n == 0) break;
            bool plain = true;
            if (stab_on && sss == stab_x && sp < last) {
                const int64_t p = skip_stable(sp, last);
                g_ff_bytes += p - sp;
                sp = p;
                if (sp == last && !eof) break;
            }
            const uint32_t x_in = sss;
            if (W->nleading && sss) {
                sss = 0;
                bool same = (sp != last) && (n == initial_count);
                for (uint32_t k = 0; same && k + 1 < n; k++) same = tidv[cur][k] == initial[k];
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
            const uint32_t c = sp < last ? in[sp] : 0u;
            if (step(sp, last)) last_matched_pos = matched[1];
            learn(plain && sp < last, c, x_in);
            if (sp == last) break;
        }
        if (last_matched_pos >= 0) {
            const int64_t p = last_matched_pos - processed;
            if (p > 0) seen_newline = in[p - 1] == '\n';
        }
        const uint32_t *ncaps = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(W) + W->multi_ncaps_off);
        if (has_matched) {
            if (matched_id >= (int64_t) W->nregexes) return -1;
            uint32_t ofs = 0;
            for (int64_t r = 0; r < matched_id; r++) ofs += ncaps[r] + 1;
            ofs *= 2;
            if (eof || n == 0) {
                const uint32_t cnt = 2u * (ncaps[matched_id] + 1);
                for (uint32_t k = 0; k < ovec_slots; k++) ov[k] = k < cnt ? matched[ofs + k] : -1;
                if (n > 0) {
                    n = 0;
                    ctx_eof = 1;
                }
                processed = matched[ofs + 1];
                empty_capture = matched[ofs] == matched[ofs + 1];
                has_matched = 0;
                first_buf = 1;
                return matched_id;
            }
            *has_pending = 1;
            pending[0] = matched[ofs];
            pending[1] = matched[ofs + 1];
        } else if (eof) {
            ctx_eof = 1;
            return -5;
        }
        processed += sp;
        if (ovec_slots >= 2) {
            /* :692-735: the range any listed thread's match could still span (end offset read without
             * the per-regex offset, :721) */
            int64_t a0 = -1, a1 = -1;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t ofs = 0;
                for (uint32_t r = 0; r < W->nregexes; r++) {
                    int64_t b = caps[cur][ofs * 64 + i];
                    if (b != -1 && (a0 == -1 || b < a0)) a0 = b;
                    b = caps[cur][1 * 64 + i];
                    if (b != -1 && (a1 == -1 || b > a1)) a1 = b;
                    ofs += 2 * (ncaps[r] + 1);
                }
            }
            ov[0] = a0;
            ov[1] = a1;
        }
        return -2;
    }

    /* one byte step at sp (sre_vm_pike.c:312-581); returns whether a MATCH was reached */
    bool step(int64_t sp, int64_t last)
    {
        const bool     at_end = sp == last;
        const uint32_t c = at_end ? 0u : in[sp];
        const uint32_t nxt = cur ^ 1u;
        const int64_t  pos1 = processed + sp + 1;
        uint32_t       m = 64, d = 64;
        uint64_t       src = 0;
        sre_pwave_list_t Ls[64];
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t t = tidv[cur][i];
            if (W->tid_match[t]) {
                if (m == 64) m = i;
                continue;
            }
            if (at_end || !((W->accept[t][c >> 5] >> (c & 31)) & 1)) continue;
            src |= 1ull << i;
            Ls[i] = lists[(uint32_t) W->tid_list[t] * SRE_PWAVE_NCTX + (c == '\n' ? 1u : 0u)];
        }
        if (m < 64) src &= (1ull << m) - 1;
        for (uint32_t i = 0; i < 64; i++) {
            if (((src >> i) & 1) && Ls[i].done) {
                d = i;
                break;
            }
        }
        if (d < 63) src &= (2ull << d) - 1;
        stamp_cur++;
        uint32_t nn = 0;
        bool     done = false, moved = false;
        for (uint32_t i = 0; i < 64; i++) {
            if (!((src >> i) & 1)) continue;
            const sre_pwave_list_t L = Ls[i];
            if (L.sss) sss = 1;
            for (uint32_t k = 0; k < L.len; k++) {
                const sre_pwave_entry_t e = ents[L.off + k];
                if (stamp[e.tid] == stamp_cur) continue;
                stamp[e.tid] = stamp_cur;
                tidv[nxt][nn] = e.tid;
                for (uint32_t s = 0; s < nslots; s++) caps[nxt][s * 64 + nn] = ((e.saves >> s) & 1) ? pos1 : caps[cur][s * 64 + i];
                /* a thread that keeps its place and its column: nothing saved on the way, and the parent's
                 * column is known to equal the one the place held (labels: copies of one column) */
                lab[nxt][nn] = e.saves ? ((stamp_cur << 6) | nn) : lab[cur][i];
                if (e.saves != 0 || nn >= n || tidv[cur][nn] != e.tid || lab[cur][nn] != lab[cur][i]) moved = true;
                nn++;
            }
            if (i == d) {
                const sre_pwave_entry_t me = ents[L.off + L.len];
                for (uint32_t s = 0; s < nslots; s++) matched[s] = ((me.saves >> s) & 1) ? pos1 : caps[cur][s * 64 + i];
                matched_id = (int64_t) W->tid_match[me.tid] - 1;
                done = true;
            }
        }
        if (!done && m < 64) {
            for (uint32_t s = 0; s < nslots; s++) matched[s] = caps[cur][s * 64 + m];
            matched_id = (int64_t) W->tid_match[tidv[cur][m]] - 1;
            done = true;
        }
        if (done) has_matched = 1;
        step_stable = !at_end && !done && !moved && nn == n;
        cur = nxt;
        n = nn;
        return done;
    }
};

}  // namespace

extern "C" {

/* the stable-run fast-forward on / off (the tests compare both with the oracle), and how many bytes it skipped */
void pwave_sim_set_ff(int on) { g_ff_enabled = on; }
int64_t pwave_sim_ff_bytes(void) { return g_ff_bytes; }

void *pwave_sim_build(const sre_program_t *prog) { return sre_pwave_build(prog); }
void pwave_sim_free(void *h) { free(h); }

/* one exec of a fresh context with eof; returns rc, fills ov[ovec_slots] and *poisoned */
int64_t pwave_sim_exec(void *h, const uint8_t *data, int64_t n, int64_t *ov, uint32_t ovec_slots, int *poisoned)
{
    const sre_pwave_hdr_t *W = static_cast<const sre_pwave_hdr_t *>(h);
    Sim s;
    s.W = W;
    s.lists = reinterpret_cast<const sre_pwave_list_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_lists);
    s.ents = reinterpret_cast<const sre_pwave_entry_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_entries);
    s.in = data;
    s.nslots = W->nslots;
    s.caps[0].assign((size_t) W->nslots * 64, -1);
    s.caps[1].assign((size_t) W->nslots * 64, -1);
    s.matched.assign(W->nslots, -1);
    const int64_t rc = s.exec(n, 0, false, ov, ovec_slots);
    if (poisoned) *poisoned = (int) s.poisoned;
    return rc;
}

/* a context fed in chunks */
void *pwave_sim_ctx_new(void *h)
{
    const sre_pwave_hdr_t *W = static_cast<const sre_pwave_hdr_t *>(h);
    Sim *s = new Sim();
    s->W = W;
    s->lists = reinterpret_cast<const sre_pwave_list_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_lists);
    s->ents = reinterpret_cast<const sre_pwave_entry_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_entries);
    s->nslots = W->nslots;
    s->caps[0].assign((size_t) W->nslots * 64, -1);
    s->caps[1].assign((size_t) W->nslots * 64, -1);
    s->matched.assign(W->nslots, -1);
    return s;
}
void pwave_sim_ctx_free(void *c) { delete static_cast<Sim *>(c); }
int64_t pwave_sim_ctx_exec(void *c, const uint8_t *data, int64_t n, int eof, int64_t *ov, uint32_t ovec_slots,
                           int *has_pending, int64_t *pending)
{
    Sim *s = static_cast<Sim *>(c);
    s->in = data;
    return s->exec_chunk(n, eof != 0, ov, ovec_slots, has_pending, pending);
}

/* the find-all iteration of sre_k_pike_scan_wave: rec[0] = last rc / final error, rec[1] = count, rec[2..] = last ovector */
void pwave_sim_count(void *h, const uint8_t *data, int64_t n, int64_t *rec, uint32_t ovec_slots)
{
    const sre_pwave_hdr_t *W = static_cast<const sre_pwave_hdr_t *>(h);
    Sim s;
    s.W = W;
    s.lists = reinterpret_cast<const sre_pwave_list_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_lists);
    s.ents = reinterpret_cast<const sre_pwave_entry_t *>(reinterpret_cast<const uint8_t *>(W) + W->off_entries);
    s.nslots = W->nslots;
    s.caps[0].assign((size_t) W->nslots * 64, -1);
    s.caps[1].assign((size_t) W->nslots * 64, -1);
    s.matched.assign(W->nslots, -1);
    int64_t off = 0, count = 0, rc, last_rc = -5;
    for (uint32_t k = 0; k < ovec_slots; k++) rec[2 + k] = -1;
    for (;;) {
        s.in = data + off;
        rc = s.exec(n - off, 0, false, rec + 2, ovec_slots);
        if (rc < 0) break;
        count++;
        last_rc = rc;
        off = s.processed;
    }
    rec[0] = rc == -1 ? rc : (count > 0 ? last_rc : rc);
    rec[1] = count;
}

}
