/*
 * sre_scan_host.cpp — device tables of the table-driven scanner: the step
 * automaton (sre_dfa.cpp) flattened for one scan mode and uploaded once per
 * scanner.
 */
#include "sre_scan_host.h"
#include "sre_scan_fast.h"
#include "sre_hip_runtime.h"
#include <sregex_hip.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>

template <typename T>
static hipError_t
upload(const std::vector<T> &v, const T **out, std::vector<void *> &owned)
{
    void      *d = NULL;
    size_t     bytes = (v.size() ? v.size() : 1) * sizeof(T);
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) return e;
    owned.push_back(d);
    if (v.size()) {
        e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    *out = static_cast<const T *>(d);
    return hipSuccess;
}

void
sre_scan_tables_release(sre_scan_device_tables_t *t)
{
    if (t == NULL) return;
    for (void *p : t->owned) (void) hipFree(p);
    if (t->d_tab) (void) hipFree(t->d_tab);
    delete t;
}

sre_scan_device_tables_t *
sre_scan_tables_build(const sre_program_t *prog, const sre_dfa_t *d, int mode, const char **why)
{
    static const char *dummy;
    if (why == NULL) why = &dummy;
    *why = NULL;
    {
        const uint32_t b = d->ncls <= 2 ? 1 : d->ncls <= 4 ? 2 : d->ncls <= 16 ? 4 : 8;
        if (d->nstates + 1 > SRE_SCAN_MAX_ROWS(b)) {
            *why = "automaton has more states than the LDS fast table holds";
            return NULL;
        }
    }
    if (d->max_threads > 254) {
        *why = "thread lists longer than 254";
        return NULL;
    }
    if (mode == SRE_HIP_PIKE_COUNT && d->has_caret && prog->nregexes > 1) {
        /* the newline flag of a re-armed context comes from slot 1, i.e. from
         * regex 0's group 0 whatever regex matched (sre_vm_pike.c:586-601) */
        *why = "COUNT with ^ or \\A over several regexes needs the per-context newline flag";
        return NULL;
    }
    if (mode == SRE_HIP_PIKE_COUNT && d->has_lookahead && prog->nregexes > 1) {
        /* ... and so does the word flag (:594) */
        *why = "COUNT with $ \\z \\b \\B over several regexes needs the per-context word flag";
        return NULL;
    }

    const uint32_t nsym = d->ncls + 1;
    sre_scan_device_tables_t *t = new sre_scan_device_tables_t();
    sre_scan_tables_t        &h = t->h;
    memset(&h, 0, sizeof(h));
    h.nstates = d->nstates;
    h.ncls = d->ncls;
    h.nslots = d->nslots;
    h.max_threads = d->max_threads;
    static_assert(SRE_SCAN_NINIT == SRE_DFA_NINIT, "initial lists");
    for (int v = 0; v < SRE_DFA_NINIT; v++) h.init[v] = d->init[v];
    h.word_restart = d->init[SRE_DFA_INIT_RESTART_WORD] != d->init[SRE_DFA_INIT_RESTART];
    h.mode = mode;
    h.fast_bytes = d->nstates * SRE_FAST_ROW_BYTES;
    h.nregexes = prog->nregexes;

    /* fast table: `stride` bytes per lookup through packed byte classes, the COUNT folds (sre_scan_fast.cpp) */
    sre_scan_fast_t F;
    sre_scan_fast_build(d, mode, &F);
    const uint32_t bits = F.bits, stride = F.stride;
    h.stride = stride;
    h.class_bits = bits;
    h.any_fresh = F.any_fresh;
    const std::vector<uint8_t> &fresh = F.fresh;
    std::vector<uint32_t>      &fast = F.fast, &fast_plain = F.fast_plain;

    /* STABLE steps (sre_hip_scan.h).  Per state s and byte class k: the step is a
     * self-loop without an event, and nm1 = the threads that descend from themselves in
     * it, saving nothing and not being the ".*?" restart.  The state's neutral set is the
     * largest nm1; a class is stable for s when its nm1 covers that set, and a fast-table
     * entry is STABLE when every byte of it is a stable class of s — so inside a stable
     * stretch the state is s at EVERY byte, not just at entry boundaries. */
    std::vector<uint16_t> neutral(d->nstates, 0);
    std::vector<uint32_t> nstable(d->nstates, 0);
    for (uint32_t s0 = 1; s0 < d->nstates; s0++) {
        const uint32_t n = d->nthreads[s0];
        if (n == 0 || n > 16) continue;
        std::vector<uint16_t> nm1(d->ncls, 0);
        for (uint32_t k = 0; k < d->ncls; k++) {
            const sre_dfa_trans_t &tr = d->t(s0, k);
            if (tr.next != s0 || tr.ev_kind != SRE_DFA_EV_NONE || tr.skipped || tr.lin_n != n) continue;
            uint16_t m = 0;
            for (uint32_t j = 0; j < n; j++) {
                const bool saves = (d->lin_saves[tr.lin_off + j] | d->lin_early[tr.lin_off + j]) != 0;
                const bool restart = d->list_pcs[d->list_off[s0] + j] == 1;
                if (d->lin_parent[tr.lin_off + j] == j && !saves && !restart) m |= (uint16_t) (1u << j);
            }
            nm1[k] = m;
            if (__builtin_popcount(m) > __builtin_popcount(neutral[s0])) neutral[s0] = m;
        }
        if (neutral[s0] == 0) continue;
        for (unsigned idx = 0; idx < 256; idx++) {
            const uint32_t e = fast_plain[(size_t) s0 * 256 + idx];
            if (e & SRE_FAST_SLOW) continue;
            bool ok = true;
            for (uint32_t sub = 0; sub < stride && ok; sub++) {
                const uint32_t k = bits == 8 ? d->cls_map[idx] : ((idx >> (sub * bits)) & ((1u << bits) - 1));
                ok = k < d->ncls && (nm1[k] & neutral[s0]) == neutral[s0];
            }
            if (!ok) continue;
            fast_plain[(size_t) s0 * 256 + idx] |= SRE_FAST_STABLE;
            if (mode != SRE_HIP_PIKE_COUNT) fast[(size_t) s0 * 256 + idx] |= SRE_FAST_STABLE;
            nstable[s0]++;
        }
    }
    /* shadow rows for the states with the most STABLE entries, as LDS allows */
    h.nshadow = 0;
    if (mode != SRE_HIP_PIKE_COUNT && getenv("SRE_HIP_NO_SHADOW") == NULL) {      /* (experiment knob) */
        const uint32_t room = SRE_SCAN_MAX_ROWS(bits) - d->nstates - 1;
        while (h.nshadow < SRE_SCAN_MAX_SHADOWS && h.nshadow < room) {
            uint32_t best = 0;
            for (uint32_t s0 = 1; s0 < d->nstates; s0++) {
                bool taken = false;
                for (uint32_t q = 0; q < h.nshadow; q++) taken |= h.shadow_state[q] == s0;
                if (!taken && nstable[s0] > nstable[best]) best = s0;
            }
            if (best == 0) break;
            h.shadow_state[h.nshadow++] = (uint8_t) best;
        }
    }
    h.fast_rows = d->nstates + 1 + h.nshadow;       /* rows of the scan kernel's LDS copy */
    h.wide = bits <= 2;
    if (mode == SRE_HIP_PIKE_COUNT && bits == 4) {
        /* one op less per lookup (-2.6 % on configs[2]) — if the larger tile still lets three
         * workgroups share a CU, which is what this mode's registers allow and what pays more
         * (configs[2]: 1.31 ms at two per CU either way, 1.08 ms narrow at three) */
        h.wide = 1;
        if (sre_scan_lds_bytes(&h) + 2 * 1024 > 160 * 1024 / 3) h.wide = 0;
        if (getenv("SRE_HIP_NO_WIDE4")) h.wide = 0;     /* experiment knob */
    }

    std::vector<sre_dev_trans_t> trans(d->trans.size());
    for (size_t i = 0; i < d->trans.size(); i++) {
        const sre_dfa_trans_t &a = d->trans[i];
        sre_dev_trans_t       &b = trans[i];
        memset(&b, 0, sizeof(b));
        b.next = a.next;
        b.kind = a.ev_kind == SRE_DFA_EV_DONE ? (a.ev_empty ? SRE_DEV_EV_DONE_EMPTY : SRE_DFA_EV_DONE)
               : a.ev_kind == SRE_DFA_EV_POP ? (a.ev_empty ? SRE_DFA_EV_POP : SRE_DEV_EV_POP_FULL) : 0;
        b.src = a.ev_src;
        b.regex = a.ev_regex;
        b.lin_off = a.lin_off;
        b.lin_n = a.lin_n;
        b.skipped = a.skipped;
        b.saves = a.ev_saves;
        b.early = a.ev_early;
    }
    std::vector<uint16_t> trans2((size_t) d->nstates * nsym);
    for (size_t i = 0; i < trans2.size(); i++) trans2[i] = (uint16_t) (trans[i].next | ((uint32_t) trans[i].kind << 8));
    std::vector<uint8_t> cls(d->cls_map, d->cls_map + 256);
    std::vector<uint8_t> flags(d->nstates);
    for (uint32_t s = 0; s < d->nstates; s++) {
        flags[s] = (uint8_t) ((d->matched[s] ? 1 : 0) | (d->seen_start[s] << 1) | (fresh[s] ? SRE_STATE_FRESH : 0));
    }
    std::vector<uint32_t> ncaps(prog->multi_ncaps, prog->multi_ncaps + prog->nregexes);
    std::vector<uint8_t>  unskip(d->nstates);
    for (uint32_t s = 0; s < d->nstates; s++) unskip[s] = (uint8_t) d->unskip[s];
    /* Lineage vectors, stored once per distinct content.  Per new thread of a
     * transition: parent index, SAVE masks, and flag bits (bit0 its closure path
     * saved a slot, bit1 it is the ".*?" ANY thread).  Equal lin_off therefore
     * means equal lineage function, which sre_k_lineage_maps uses to skip runs
     * of an idempotent one (a thread list looping in place: x+ over a long run
     * of x). */
    std::vector<uint8_t>  lin_parent, lin_flags;
    std::vector<uint64_t> lin_saves, lin_early;
    {
        std::map<std::vector<uint64_t>, uint32_t> canon;
        for (size_t i = 0; i < d->trans.size(); i++) {
            const sre_dfa_trans_t &a = d->trans[i];
            std::vector<uint64_t>  key;
            std::vector<uint8_t>   fl(a.lin_n);
            key.push_back(a.lin_n);
            for (uint32_t j = 0; j < a.lin_n; j++) {
                uint8_t f = (d->lin_saves[a.lin_off + j] | d->lin_early[a.lin_off + j]) ? 1 : 0;
                if (a.next != SRE_DFA_DEAD && d->list_pcs[d->list_off[a.next] + j] == 1) f |= 2;
                fl[j] = f;
                key.push_back(d->lin_parent[a.lin_off + j] | ((uint64_t) f << 8));
                key.push_back(d->lin_saves[a.lin_off + j]);
                key.push_back(d->lin_early[a.lin_off + j]);
            }
            auto it = canon.find(key);
            if (it == canon.end()) {
                it = canon.emplace(key, (uint32_t) lin_parent.size()).first;
                for (uint32_t j = 0; j < a.lin_n; j++) {
                    lin_parent.push_back(d->lin_parent[a.lin_off + j]);
                    lin_saves.push_back(d->lin_saves[a.lin_off + j]);
                    lin_early.push_back(d->lin_early[a.lin_off + j]);
                    lin_flags.push_back(fl[j]);
                }
            }
            trans[i].lin_off = it->second;

            /* idempotent: applying the map twice == once, for every input.  As
             * (source, saved-constant, stop-constant) per entry; a source of
             * NONE forces stop, and then the ancestor nibble is never used. */
            const uint32_t off = it->second;
            bool           idem = a.lin_n > 0 && a.lin_n <= 16;
            for (uint32_t j = 0; idem && j < a.lin_n; j++) {
                const uint32_t p1 = lin_parent[off + j];
                const bool     c1 = (lin_flags[off + j] & 1) != 0, c2 = (lin_flags[off + j] & 2) != 0;
                if (p1 == SRE_DFA_NO_PARENT) continue;          /* constant either way */
                if (p1 >= a.lin_n) {
                    idem = false;                               /* refers to a thread the new list lacks */
                    break;
                }
                const uint32_t p2 = lin_parent[off + p1];
                const bool     d1 = (lin_flags[off + p1] & 1) != 0, d2 = (lin_flags[off + p1] & 2) != 0;
                /* twice: source p2 (through p1), constants c | d */
                const bool stop_twice = c2 || d2 || p2 == SRE_DFA_NO_PARENT;
                if (stop_twice != c2) idem = false;             /* once: stop = stop_in[p1] | c2 — differs unless forced equal */
                if (!stop_twice && p2 != p1) idem = false;
                if ((c1 || d1) != c1) idem = false;
            }
            trans[i].pad = idem ? 1 : 0;
        }
    }
    h.lin_total = (uint32_t) lin_parent.size();
    h.list_total = (uint32_t) d->list_pcs.size();

    {
        /* the scan kernel addresses its fast table (first in dynamic LDS, behind
         * at most 8.5 KiB of static LDS) with 16-bit LDS addresses and may ask
         * for SRE_SCAN_LDS_LIMIT in all; the capture walker and the lineage kernel
         * may take SRE_CAPTURE_LDS_LIMIT.  Otherwise the exact VM engine takes the
         * program. */
        const size_t tr = ((size_t) d->nstates * nsym + SRE_SCAN_NINIT) * sizeof(sre_dev_trans_t);
        const size_t scan_lds = sre_scan_lds_bytes(&h);
        const size_t fast_end = (size_t) h.fast_rows * SRE_FAST_ROW_BYTES + 768 + (bits == 8 ? 512 : (8 / bits) * 512);
        const size_t cap_lds = (size_t) h.fast_bytes + 256 + tr + (size_t) h.lin_total * 9
                               + ((size_t) d->nstates + 1 + h.list_total) * 4 + 16 + 512;
        const size_t lin_lds = 256 + (size_t) d->nstates * nsym * sizeof(sre_dev_trans_t)
                               + 2 * (size_t) ((h.lin_total + 15u) & ~15u) + 512;
        if (scan_lds > SRE_SCAN_LDS_LIMIT || fast_end > 64 * 1024 || cap_lds > SRE_CAPTURE_LDS_LIMIT
            || lin_lds > SRE_CAPTURE_LDS_LIMIT)
        {
            *why = "automaton tables exceed the LDS budget of the scanner";
            delete t;
            return NULL;
        }
    }

    hipError_t e;
    if ((e = upload(fast, &h.fast, t->owned)) != hipSuccess
        || (e = upload(fast_plain, &h.fast_plain, t->owned)) != hipSuccess
        || (e = upload(cls, &h.cls, t->owned)) != hipSuccess
        || (e = upload(trans, &h.trans, t->owned)) != hipSuccess
        || (e = upload(trans2, &h.trans2, t->owned)) != hipSuccess
        || (e = upload(lin_parent, &h.lin_parent, t->owned)) != hipSuccess
        || (e = upload(lin_saves, &h.lin_saves, t->owned)) != hipSuccess
        || (d->has_lookahead && (e = upload(lin_early, &h.lin_early, t->owned)) != hipSuccess)
        || (e = upload(lin_flags, &h.lin_flags, t->owned)) != hipSuccess
        || (e = upload(flags, &h.state_flags, t->owned)) != hipSuccess
        || (e = upload(d->list_off, &h.list_off, t->owned)) != hipSuccess
        || (e = upload(d->list_pcs, &h.list_pcs, t->owned)) != hipSuccess
        || (e = upload(ncaps, &h.multi_ncaps, t->owned)) != hipSuccess
        || (e = upload(neutral, &h.neutral, t->owned)) != hipSuccess
        || (e = upload(unskip, &h.unskip, t->owned)) != hipSuccess
        || (e = hipMalloc(reinterpret_cast<void **>(&t->d_tab), sizeof(h))) != hipSuccess
        || (e = hipMemcpy(t->d_tab, &h, sizeof(h), hipMemcpyHostToDevice)) != hipSuccess)
    {
        sre_hip_fail("scanner table upload", e);
        sre_scan_tables_release(t);
        *why = "device allocation failed";
        return NULL;
    }
    return t;
}
