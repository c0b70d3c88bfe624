/*
 * sre_dfa.cpp — builds the step automaton described in sre_dfa.h by running
 * the reference VM's step on capture-free thread lists.
 *
 * Mirrors (capture-free): byte loop sre_vm_pike.c:314-567, closure
 * sre_vm_pike.c:756-942.  \A and ^ are resolved at add time from the byte just
 * consumed (:839-864) and are therefore functions of the input symbol.
 *
 * Look-ahead assertions ($ \z \b \B) wait in the list and are decided by the
 * NEXT symbol (:450-504).  One that holds is replaced, within the same step, by
 * the closure of its continuation at the SAME position, de-duplicated against
 * the generation that built the current list and spliced in at the head of the
 * list (:506-526).  For a list holding such threads the state key therefore
 * also carries (i) what the byte in front of the position was (word character /
 * newline / buffer start: the threads' seen_word, and ^ \A inside a splice) and
 * (ii) the set of instructions that generation visited.  SAVEs executed by a
 * splice carry the position BEFORE the byte ("early" masks); everything else
 * about a spliced thread is folded into the thread of the list it came from.
 *
 * The leading-byte skip (sre_vm_pike.c:256-309) is part of the observable
 * behaviour (its initial-state test ignores the last thread, :266-273, so it
 * can re-seed a search that already holds a match) and is modelled byte by
 * byte: a state that passes that test moves, on a non-leading byte, to the
 * freshly seeded list.  For that the state key also carries the
 * seen_start_state flag and which initial list the search was seeded from.
 */
#include "sre_dfa.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <deque>
#include <map>

namespace {

enum { PREV_WORD = 1, PREV_NL = 2, PREV_START = 4 };

struct Builder {
    const sre_program_t *prog;
    sre_dfa_t           *d;
    std::vector<uint32_t> tags;
    uint32_t             gen = 0, gen_alloc = 0;
    std::map<std::vector<uint32_t>, uint32_t> ids;     /* key: pcs..., flags [, prev, visited pcs...] */
    std::vector<std::vector<uint32_t>> lists;          /* per state */
    std::vector<uint8_t>               sss, variant;   /* per state */
    std::vector<uint8_t>               prevk;          /* per state: PREV_* of the byte in front, as a SPLICE sees it
                                                          (^ / \A in its closure, \b / \B threads it lists and
                                                          decides in the same step) */
    std::vector<uint8_t>               prevw;          /* per state: the byte in front is a word byte, as the LISTED
                                                          \b / \B threads see it (their own seen_word).  Differs from
                                                          prevk only in the states a chunk boundary makes (rekind) */
    std::vector<std::vector<uint8_t>>  fresh;          /* per state, per thread: a look-ahead assertion (or MATCH)
                                                          thread whose own closure path saved a group-0 start,
                                                          i.e. a match it completes at this position is empty */
    uint64_t                           slot0_mask = 0; /* the group-0 start slots of all regexes */
    unsigned                           prev_mask = PREV_WORD | PREV_NL | PREV_START;   /* kinds of the byte in front
                                                          that some assertion of the program can tell apart */
    std::vector<std::vector<uint32_t>> visited;        /* per state: instructions tagged by the generation
                                                          that built the list (kept if it holds a look-ahead) */
    bool                 visited_start = false;        /* pc 0 reached by the last closure(s) */
    std::vector<uint32_t> vis_now;                     /* instructions tagged while building the new list */

    /* result of one closure / step */
    std::vector<uint32_t> nl;          /* new list pcs */
    std::vector<uint8_t>  npar;
    std::vector<uint64_t> nsav, nearly;

    bool consumes(const sre_insn_t &in, unsigned c) const
    {
        switch (in.opcode) {
        case SRE_OP_CHAR:  return c == in.ch;
        case SRE_OP_ANY:   return true;
        case SRE_OP_IN:    return sre_in_ranges(&prog->ranges[in.x], in.nranges, c) != 0;
        case SRE_OP_NOTIN: return sre_in_ranges(&prog->ranges[in.x], in.nranges, c) == 0;
        default:           return false;
        }
    }

    bool is_lookahead(uint32_t pc) const
    {
        const sre_insn_t &in = prog->insns[pc];
        return in.opcode == SRE_OP_ASSERT && (in.ch & SRE_ASSERT_LOOKAHEAD);
    }

    bool holds_lookahead(const std::vector<uint32_t> &pcs) const
    {
        for (uint32_t pc : pcs) {
            if (is_lookahead(pc)) return true;
        }
        return false;
    }

    /*
     * Capture-free closure (sre_vm_pike.c:756-942) under generation `gen`.
     * Appends to out_pc/out_par/out_sav.  Returns true when MATCH was reached
     * with from_loop set (SRE_DONE); *done_saves / *done_regex then describe
     * the match.  `track`: remember what gets tagged (vis_now).
     */
    bool closure(uint32_t pc0, bool a_ok, bool caret_ok, bool from_loop, uint8_t parent,
                 std::vector<uint32_t> &out_pc, std::vector<uint8_t> &out_par,
                 std::vector<uint64_t> &out_sav, uint64_t *done_saves, uint32_t *done_regex,
                 bool track)
    {
        struct Rec { uint32_t pc; uint64_t mask; };
        std::vector<Rec> stack;
        uint32_t pc = pc0;
        uint64_t mask = 0;

        for (;;) {
            /* one chain of tail calls */
            for (;;) {
                const sre_insn_t &in = prog->insns[pc];
                if (tags[pc] == gen) {
                    if (in.opcode == SRE_OP_SPLIT && tags[in.y] != gen) {   /* :774-784 */
                        if (pc == 0) visited_start = true;
                        pc = in.y;
                        continue;
                    }
                    break;
                }
                tags[pc] = gen;
                if (track) vis_now.push_back(pc);
                if (pc == 0) visited_start = true;                          /* :799-802 */
                if (in.opcode == SRE_OP_JMP) {
                    pc = in.x;
                    continue;
                }
                if (in.opcode == SRE_OP_SPLIT) {
                    stack.push_back(Rec{in.y, mask});
                    pc = in.x;
                    continue;
                }
                if (in.opcode == SRE_OP_SAVE) {
                    mask |= 1ull << in.arg;
                    pc++;
                    continue;
                }
                if (in.opcode == SRE_OP_ASSERT && !(in.ch & SRE_ASSERT_LOOKAHEAD)) {
                    bool ok = in.ch == SRE_ASSERT_BIG_A ? a_ok : caret_ok;   /* :839-864 */
                    if (!ok) break;
                    pc++;
                    continue;
                }
                if (in.opcode == SRE_OP_MATCH && from_loop) {
                    *done_saves = mask;
                    *done_regex = in.arg;
                    return true;
                }
                /* consuming instruction, look-ahead assertion (:866-884) or MATCH */
                out_pc.push_back(pc);
                out_par.push_back(parent);
                out_sav.push_back(mask);
                break;
            }
            if (stack.empty()) return false;
            pc = stack.back().pc;
            mask = stack.back().mask;
            stack.pop_back();
        }
    }

    uint32_t intern(const std::vector<uint32_t> &pcs, bool matched, int seen_start, int var,
                    unsigned prev, const std::vector<uint64_t> &saves)
    {
        if (pcs.empty()) return SRE_DFA_DEAD;
        std::vector<uint8_t> fr(pcs.size(), 0);
        for (size_t i = 0; i < pcs.size(); i++) {
            if ((is_lookahead(pcs[i]) || prog->insns[pcs[i]].opcode == SRE_OP_MATCH) && (saves[i] & slot0_mask)) fr[i] = 1;
        }
        std::vector<uint32_t> vis;
        if (holds_lookahead(pcs)) {
            /* what a splice out of this list will see (:506-526): the marks of
             * the generation that built the list which are STILL in place — a
             * splice later in the same step overwrites, with the older
             * generation, the marks of instructions it walks through */
            for (uint32_t pc = 0; pc < prog->len; pc++) {
                if (tags[pc] == gen) vis.push_back(pc);
            }
        }
        return intern_raw(pcs, matched, seen_start, var, prev, (prev & PREV_WORD) != 0, vis, fr);
    }

    /* seen_start: 0 clear, 1 set (consumed by the next check), 2 set by a skip
     * re-seed that is still travelling to its target byte (see step) */
    uint32_t intern_raw(const std::vector<uint32_t> &pcs, bool matched, int seen_start, int var,
                        unsigned prev, bool listed_word, const std::vector<uint32_t> &vis_in,
                        const std::vector<uint8_t> &fr)
    {
        if (pcs.empty()) return SRE_DFA_DEAD;
        if (prog->nleading == 0) {       /* the skip does not exist: flags are inert */
            seen_start = 0;
            var = 0;
        }
        if (!d->has_caret) var = 0;      /* all three initial lists coincide */
        std::vector<uint32_t> key(pcs);
        key.push_back((matched ? 1u : 0u) | ((uint32_t) seen_start << 1) | ((uint32_t) var << 3));
        std::vector<uint32_t> vis;
        if (holds_lookahead(pcs)) {
            vis = vis_in;
            prev &= prev_mask;
            if (!(prev_mask & PREV_WORD)) listed_word = false;
            key.push_back(0x80000000u | prev | (listed_word ? 0x100u : 0u));
            key.insert(key.end(), vis.begin(), vis.end());
            key.push_back(0xc0000000u);
            for (size_t i = 0; i < fr.size(); i++) {
                if (fr[i]) key.push_back((uint32_t) i);
            }
        } else {
            prev = 0;
            listed_word = false;
        }
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        uint32_t id = (uint32_t) lists.size();
        ids.emplace(key, id);
        lists.push_back(pcs);
        d->matched.push_back(matched ? 1 : 0);
        sss.push_back((uint8_t) seen_start);
        variant.push_back((uint8_t) var);
        prevk.push_back((uint8_t) prev);
        prevw.push_back(listed_word ? 1 : 0);
        visited.push_back(vis);
        fresh.push_back(fr);
        return id;
    }

    /* sre_vm_pike.c:992-1061: can byte c start a match? */
    bool is_leading(unsigned c) const
    {
        if (prog->leading_byte != -1) return (int) c == prog->leading_byte;
        for (uint32_t i = 0; i < prog->nleading; i++) {
            if (consumes(prog->insns[prog->leading_insns[i]], c)) return true;
        }
        return false;
    }
};

}  // namespace

extern "C" void
sre_dfa_free(sre_dfa_t *dfa)
{
    delete dfa;
}

extern "C" sre_dfa_t *
sre_dfa_build(const sre_program_t *prog, uint32_t max_states, const char **why)
{
    return sre_dfa_build2(prog, max_states, 0, why);
}

extern "C" sre_dfa_t *
sre_dfa_build2(const sre_program_t *prog, uint32_t max_states, int chunk_twins, const char **why)
{
    const char *dummy;
    if (why == NULL) why = &dummy;
    *why = NULL;

    if (prog->nslots > 64) {
        *why = "more than 64 capture slots";
        return NULL;
    }

    sre_dfa_t *d = new sre_dfa_t();
    Builder    b;
    b.prog = prog;
    b.d = d;
    b.tags.assign(prog->len + 1, 0);
    d->nslots = prog->nslots;
    d->has_caret = 0;
    d->has_lookahead = prog->lookahead_asserts ? 1 : 0;

    /* ---- byte classes: bytes no consuming instruction can tell apart ---- */
    {
        std::map<std::vector<uint8_t>, uint32_t> sigs;
        bool has_dollar = false, has_b = false;
        for (uint32_t pc = 0; pc < prog->len; pc++) {
            const sre_insn_t &in = prog->insns[pc];
            if (in.opcode != SRE_OP_ASSERT) continue;
            if (in.ch & (SRE_ASSERT_BIG_A | SRE_ASSERT_CARET)) d->has_caret = 1;
            if (in.ch & SRE_ASSERT_DOLLAR) has_dollar = true;
            if (in.ch & (SRE_ASSERT_SMALL_B | SRE_ASSERT_BIG_B)) has_b = true;
        }
        if (!has_b) b.prev_mask &= ~(unsigned) PREV_WORD;
        for (unsigned c = 0; c < 256; c++) {
            std::vector<uint8_t> sig;
            for (uint32_t pc = 0; pc < prog->len; pc++) {
                const sre_insn_t &in = prog->insns[pc];
                if (in.opcode == SRE_OP_CHAR || in.opcode == SRE_OP_IN || in.opcode == SRE_OP_NOTIN) {
                    sig.push_back(b.consumes(in, c) ? 1 : 0);
                }
            }
            if (d->has_caret || has_dollar) sig.push_back(c == '\n');
            if (has_b) sig.push_back(sre_isword(c) ? 1 : 0);
            auto it = sigs.find(sig);
            if (it == sigs.end()) it = sigs.emplace(sig, (uint32_t) sigs.size()).first;
            d->cls_map[c] = (uint8_t) it->second;
            if (sigs.size() > 255) {
                *why = "more than 255 byte classes";
                delete d;
                return NULL;
            }
        }
        d->ncls = (uint32_t) sigs.size();
    }
    std::vector<int> rep(d->ncls, -1);
    for (int c = 255; c >= 0; c--) rep[d->cls_map[c]] = c;

    /* ---- state 0 = DEAD; then the three initial lists ---- */
    b.lists.push_back(std::vector<uint32_t>());
    d->matched.push_back(0);
    b.sss.push_back(0);
    b.variant.push_back(0);
    b.prevk.push_back(0);
    b.prevw.push_back(0);
    b.visited.push_back(std::vector<uint32_t>());
    b.fresh.push_back(std::vector<uint8_t>());
    {
        uint32_t slot = 0;
        for (uint32_t r = 0; r < prog->nregexes; r++) {
            b.slot0_mask |= 1ull << slot;
            slot += 2 * (prog->multi_ncaps[r] + 1);
        }
    }
    std::vector<std::vector<uint32_t>> init_lists(SRE_DFA_NINIT);
    std::vector<std::vector<uint8_t>>  lin_par_of_init(SRE_DFA_NINIT);
    std::vector<std::vector<uint64_t>> lin_sav_of_init(SRE_DFA_NINIT);
    for (int v = 0; v < SRE_DFA_NINIT; v++) {
        uint64_t ds;
        uint32_t dr;
        b.gen = ++b.gen_alloc;
        b.nl.clear();
        b.npar.clear();
        b.nsav.clear();
        b.vis_now.clear();
        b.closure(0, v == SRE_DFA_INIT_START, v == SRE_DFA_INIT_START || v == SRE_DFA_INIT_RESTART_NL, false,
                  SRE_DFA_NO_PARENT, b.nl, b.npar, b.nsav, &ds, &dr, true);
        /* in front of the position: the buffer start, or the byte the context remembers
         * (seen_newline / seen_word, sre_vm_pike.c:586-601) */
        d->init[v] = b.intern(b.nl, false, 1, v == SRE_DFA_INIT_RESTART_WORD ? (int) SRE_DFA_INIT_RESTART : v,
                              v == SRE_DFA_INIT_START ? PREV_START : v == SRE_DFA_INIT_RESTART_NL ? PREV_NL
                              : v == SRE_DFA_INIT_RESTART_WORD ? PREV_WORD : 0, b.nsav);
        lin_par_of_init[v] = b.npar;
        lin_sav_of_init[v] = b.nsav;
        init_lists[v] = b.nl;
    }

    /* ---- breadth-first exploration ---- */
    const uint32_t nsym = d->ncls + 1;
    std::vector<uint32_t> unskip_of, rekind_of;
    for (uint32_t s = 0; s < b.lists.size(); s++) {
        if (b.lists.size() > max_states) {
            *why = "state cap exceeded";
            delete d;
            return NULL;
        }
        /* the chunk-boundary twin of a travelling skip (sre_dfa.h `unskip`): interned now so
         * that it is explored like any other state */
        if (unskip_of.size() <= s) unskip_of.resize(s + 1, 0);
        unskip_of[s] = s;
        if (prog->nleading && (prog->lookahead_asserts == 0 || chunk_twins) && b.sss[s] == 2) {
            /* (look-ahead programs: only in the automaton of a chunked stream) */
            const std::vector<uint32_t> Lc = b.lists[s], vis = b.visited[s];
            const std::vector<uint8_t>  fr = b.fresh[s];
            unskip_of[s] = b.intern_raw(Lc, d->matched[s] != 0, 1, b.variant[s], b.prevk[s], b.prevw[s] != 0, vis, fr);
        }
        if (chunk_twins) {
            /* what the context's flags make of the byte in front (sre_dfa.h `rekind`): a splice at
             * the first byte of a chunk runs at pos == 0, where ^ goes by seen_newline (:851-860)
             * and a \b / \B thread it lists starts from seen_word == 0 (:866-880); a LISTED \b / \B
             * thread kept its own seen_word, and at sp == input the context's flag is OR-ed to
             * either (:472-473, 492) */
            if (rekind_of.size() < 4 * (size_t) (s + 1)) rekind_of.resize(4 * (size_t) (s + 1), 0);
            const std::vector<uint32_t> Lc = b.lists[s], vis = b.visited[s];
            const std::vector<uint8_t>  fr = b.fresh[s];
            for (unsigned f = 0; f < 4; f++) {
                /* f == 3: the Thompson VM, whose \A / ^ / \b hold "at the start of the buffer" of
                 * every call (sre_vm_thompson.c:302-317, 320-325) and which has no context flags */
                rekind_of[4 * (size_t) s + f] =
                    b.holds_lookahead(Lc) ? b.intern_raw(Lc, d->matched[s] != 0, b.sss[s], b.variant[s],
                                                         f == 1 ? PREV_NL : f == 2 ? PREV_WORD : f == 3 ? PREV_START : 0,
                                                         b.prevw[s] != 0 || f == 2, vis, fr)
                                          : s;
            }
        }
        const std::vector<uint32_t> L = b.lists[s];
        const bool                  was_matched = d->matched[s] != 0;
        if (L.size() > d->max_threads) d->max_threads = (uint32_t) L.size();

        for (uint32_t sym = 0; sym < nsym; sym++) {
            sre_dfa_trans_t t;
            memset(&t, 0, sizeof(t));
            const bool eof = (sym == d->ncls);
            const int  c = eof ? -1 : rep[sym];

            /* generation of the current list (what a splice de-duplicates
             * against), then the one the new list is built under */
            const uint32_t g_list = ++b.gen_alloc;
            for (uint32_t pc : b.visited[s]) b.tags[pc] = g_list;
            const uint32_t g_new = ++b.gen_alloc;
            b.gen = g_new;
            b.nl.clear();
            b.npar.clear();
            b.nsav.clear();
            b.nearly.clear();
            b.vis_now.clear();
            b.visited_start = false;
            const unsigned prev_here = b.prevk[s];
            const unsigned prev_next = eof ? 0u : ((sre_isword((unsigned) c) ? PREV_WORD : 0u)
                                                   | (c == '\n' ? PREV_NL : 0u));

            /* :256-309.  In the reference one check can jump sp over many
             * bytes, re-seed at the target and run that byte's step in the SAME
             * iteration, i.e. without a second check and with the flag the
             * re-seed has just set.  Byte by byte that is: flag value 2 =
             * "travelling": keep re-seeding on non-leading bytes, and on the
             * target byte step normally with the flag still set.  Flag value 1
             * is consumed by a check, which skips only when the list "equals"
             * the search's initial closure — a comparison that leaves out the
             * last thread of both lists (:266-273). */
            bool skip = false;
            int  base_flag = 0;
            const int var = b.variant[s];
            if (prog->nleading && b.sss[s] == 2) {
                skip = !eof && !b.is_leading((unsigned) c);
                base_flag = 1;
            } else if (prog->nleading && b.sss[s] == 1 && !eof) {
                const std::vector<uint32_t> &I = init_lists[var];
                skip = (L.size() == I.size());
                for (size_t i = 0; skip && i + 1 < L.size(); i++) {
                    if (L[i] != I[i]) skip = false;
                }
                if (skip && b.is_leading((unsigned) c)) skip = false;    /* p == sp */
            }
            if (skip) {
                /* re-seed one byte further with a fresh capture (:286-302) */
                uint64_t ds;
                uint32_t dr;
                b.closure(0, false, c == '\n', false, SRE_DFA_NO_PARENT, b.nl, b.npar, b.nsav, &ds, &dr, true);
                b.nearly.assign(b.nl.size(), 0);
                t.skipped = 1;
                t.next = b.intern(b.nl, was_matched, 2, var, prev_next, b.nsav);
            } else {
                /* the list as a work queue: a look-ahead assertion that holds
                 * puts the closure of its continuation in front (:506-526) */
                struct Item { uint32_t pc; uint8_t src; uint64_t early; bool spliced; };
                std::deque<Item> work;
                for (size_t idx = 0; idx < L.size(); idx++) work.push_back(Item{L[idx], (uint8_t) idx, 0, false});
                const bool listed_word = b.prevw[s] != 0;
                /* A look-ahead assertion inside an empty loop can make the splice
                 * re-mark and re-list in a cycle; the reference VM then duplicates
                 * threads without bound (and crashes).  No automaton for that. */
                size_t budget = 64 * ((size_t) prog->len + 16);
                while (!work.empty()) {
                    if (budget-- == 0) {
                        *why = "assertion splice does not terminate (the reference VM diverges on this program)";
                        delete d;
                        return NULL;
                    }
                    const Item it = work.front();
                    work.pop_front();
                    const sre_insn_t &in = prog->insns[it.pc];
                    if (in.opcode == SRE_OP_MATCH) {                  /* :530-553 */
                        t.ev_kind = SRE_DFA_EV_POP;
                        t.ev_src = it.src;
                        t.ev_regex = (uint16_t) in.arg;
                        t.ev_early = it.early;
                        /* its group-0 start was saved at this very position: by the splice
                         * that listed it, or on the closure path of the thread it descends from */
                        t.ev_empty = ((it.early & b.slot0_mask) != 0 || b.fresh[s][it.src]) ? 1 : 0;
                        break;
                    }
                    if (in.opcode == SRE_OP_ASSERT) {                 /* :450-504 */
                        bool hold = false;
                        const bool word_here = !eof && sre_isword((unsigned) c);
                        switch (in.ch) {
                        case SRE_ASSERT_SMALL_Z: hold = eof; break;
                        case SRE_ASSERT_DOLLAR:  hold = eof || c == '\n'; break;
                        case SRE_ASSERT_SMALL_B:
                            hold = (it.spliced ? (prev_here & PREV_WORD) != 0 : listed_word) != word_here;
                            break;
                        case SRE_ASSERT_BIG_B:
                            hold = (it.spliced ? (prev_here & PREV_WORD) != 0 : listed_word) == word_here;
                            break;
                        default: break;
                        }
                        if (!hold) continue;
                        std::vector<uint32_t> sp_pc;
                        std::vector<uint8_t>  sp_par;
                        std::vector<uint64_t> sp_sav;
                        uint64_t ds = 0;
                        uint32_t dr = 0;
                        b.gen = g_list;                               /* ctx->tag-- */
                        b.closure(it.pc + 1, (prev_here & PREV_START) != 0,
                                  (prev_here & (PREV_START | PREV_NL)) != 0, false, it.src,
                                  sp_pc, sp_par, sp_sav, &ds, &dr, false);
                        b.gen = g_new;                                /* ctx->tag++ */
                        for (size_t k = sp_pc.size(); k-- > 0;) {
                            work.push_front(Item{sp_pc[k], it.src, it.early | sp_sav[k], true});
                        }
                        continue;
                    }
                    if (eof || !b.consumes(in, (unsigned) c)) continue;
                    uint64_t     ds = 0;
                    uint32_t     dr = 0;
                    const size_t before = b.nl.size();
                    const bool   done = b.closure(it.pc + 1, false, c == '\n', true, it.src,
                                                  b.nl, b.npar, b.nsav, &ds, &dr, true);
                    for (size_t k = before; k < b.nl.size(); k++) b.nearly.push_back(it.early & ~b.nsav[k]);
                    if (done) {
                        t.ev_kind = SRE_DFA_EV_DONE;                  /* :356-358, 535-553 */
                        t.ev_src = it.src;
                        t.ev_regex = (uint16_t) dr;
                        t.ev_saves = ds;
                        t.ev_early = it.early & ~ds;
                        t.ev_empty = (ds & b.slot0_mask) ? 1 : 0;     /* start == end == pos + 1 */
                        break;
                    }
                }
                t.next = b.intern(b.nl, was_matched || t.ev_kind != SRE_DFA_EV_NONE,
                                  (base_flag || b.visited_start) ? 1 : 0, var, prev_next, b.nsav);
            }
            t.lin_off = (uint32_t) d->lin_parent.size();
            t.lin_n = (uint16_t) b.nl.size();
            d->lin_parent.insert(d->lin_parent.end(), b.npar.begin(), b.npar.end());
            d->lin_saves.insert(d->lin_saves.end(), b.nsav.begin(), b.nsav.end());
            d->lin_early.insert(d->lin_early.end(), b.nearly.begin(), b.nearly.end());
            d->trans.push_back(t);
        }
    }

    d->nstates = (uint32_t) b.lists.size();
    unskip_of.resize(d->nstates);
    for (uint32_t s = 0; s < d->nstates; s++) {
        if (unskip_of[s] == 0 && s != 0) unskip_of[s] = s;     /* states interned by the last explored ones */
    }
    d->unskip = unskip_of;
    if (chunk_twins) {
        rekind_of.resize(4 * (size_t) d->nstates, 0);
        d->rekind = rekind_of;
    }
    d->seen_start = b.sss;
    d->nthreads.resize(d->nstates);
    d->list_off.resize(d->nstates + 1);
    for (uint32_t s = 0; s < d->nstates; s++) {
        d->list_off[s] = (uint32_t) d->list_pcs.size();
        d->nthreads[s] = (uint16_t) b.lists[s].size();
        d->list_pcs.insert(d->list_pcs.end(), b.lists[s].begin(), b.lists[s].end());
    }
    d->list_off[d->nstates] = (uint32_t) d->list_pcs.size();

    /* the initial lists' own SAVEs (value = search start) ride as pseudo
     * transitions appended after the real ones: trans[nstates * nsym + v] */
    for (int v = 0; v < SRE_DFA_NINIT; v++) {
        sre_dfa_trans_t t;
        memset(&t, 0, sizeof(t));
        t.next = d->init[v];
        t.lin_off = (uint32_t) d->lin_parent.size();
        t.lin_n = (uint16_t) lin_par_of_init[v].size();
        d->lin_parent.insert(d->lin_parent.end(), lin_par_of_init[v].begin(), lin_par_of_init[v].end());
        d->lin_saves.insert(d->lin_saves.end(), lin_sav_of_init[v].begin(), lin_sav_of_init[v].end());
        d->lin_early.insert(d->lin_early.end(), lin_sav_of_init[v].size(), 0);
        d->trans.push_back(t);
    }
    return d;
}
