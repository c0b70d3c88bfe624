/*
 * sre_nfa.cpp — builds the bit-parallel form described in sre_nfa.h.
 *
 * The closure mirrors sre_vm_pike.c:756-942 / sre_vm_thompson.c:273-345 on sets:
 * JMP and SPLIT are followed, SAVE is skipped, ^ and \A are decided from the
 * byte just consumed, every list-able instruction reached becomes a member.
 * (The Pike SPLIT re-descent, :774-784, changes the ORDER in which members are
 * listed, never the set.)
 */
#include "sre_nfa.h"
#include <string.h>
#include <set>

namespace {

struct NfaBuilder {
    const sre_program_t *prog;

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

    /* list-able instructions reachable from pc0 through epsilon edges */
    void closure(uint32_t pc0, bool a_ok, bool caret_ok, std::set<uint32_t> &out) const
    {
        std::vector<uint32_t> stack(1, pc0);
        std::vector<uint8_t>  seen(prog->len + 1, 0);
        while (!stack.empty()) {
            uint32_t pc = stack.back();
            stack.pop_back();
            while (pc < prog->len && !seen[pc]) {
                seen[pc] = 1;
                const sre_insn_t &in = prog->insns[pc];
                if (in.opcode == SRE_OP_JMP) {
                    pc = in.x;
                } else if (in.opcode == SRE_OP_SPLIT) {
                    stack.push_back(in.y);
                    pc = in.x;
                } else if (in.opcode == SRE_OP_SAVE) {
                    pc++;
                } else if (in.opcode == SRE_OP_ASSERT && !(in.ch & SRE_ASSERT_LOOKAHEAD)) {
                    if (!(in.ch == SRE_ASSERT_BIG_A ? a_ok : caret_ok)) break;     /* :839-864 */
                    pc++;
                } else {
                    /* consuming instruction, MATCH, or a look-ahead assertion: listed (:866-884) */
                    out.insert(pc);
                    break;
                }
            }
        }
    }
};

}  // namespace

extern "C" void
sre_nfa_free(sre_nfa_t *nfa)
{
    delete nfa;
}

extern "C" sre_nfa_t *
sre_nfa_build(const sre_program_t *prog, const char **why)
{
    static const char *dummy;
    if (why == NULL) why = &dummy;
    *why = NULL;
    if (prog->lookahead_asserts > 8) {
        *why = "more than 8 look-ahead assertions ($ \\z \\b \\B)";
        return NULL;
    }
    if (prog->nthreads > SRE_NFA_MAX_BITS || prog->len > 4096) {
        *why = "more than 64 list-able threads";
        return NULL;
    }
    NfaBuilder b;
    b.prog = prog;

    /* pc-level follow sets, without and with ^ true */
    std::vector<std::set<uint32_t>> fol[2];
    fol[0].resize(prog->len);
    fol[1].resize(prog->len);
    std::vector<uint32_t> listable, asserts;
    for (uint32_t pc = 0; pc < prog->len; pc++) {
        const sre_insn_t &in = prog->insns[pc];
        switch (in.opcode) {
        case SRE_OP_CHAR: case SRE_OP_IN: case SRE_OP_NOTIN: case SRE_OP_ANY:
            b.closure(pc + 1, false, false, fol[0][pc]);
            b.closure(pc + 1, false, true, fol[1][pc]);
            listable.push_back(pc);
            break;
        case SRE_OP_MATCH:
            listable.push_back(pc);
            break;
        case SRE_OP_ASSERT:
            if (in.ch & SRE_ASSERT_LOOKAHEAD) asserts.push_back(pc);
            break;
        default:
            break;
        }
    }

    /* A look-ahead assertion inside a LOOP: what a splice lists there is decided by the VM's
     * generation tags (a splice later in the same step walks through instructions an earlier
     * one has marked, sre_vm_pike.c:506-526, :770-792) — sets have no such memory, the
     * expansion tables would list threads the VM drops (found by the CPU model on random
     * patterns: `$(?:x*\n?|\B)*?x`, `(?:b*?c?|\B)+?\s??\z.`).  Such programs keep the exact VM. */
    for (uint32_t a : asserts) {
        std::vector<uint8_t>  seen(prog->len + 1, 0);
        std::vector<uint32_t> stack(1, a + 1);
        bool                  loops = false;
        while (!stack.empty() && !loops) {
            const uint32_t pc = stack.back();
            stack.pop_back();
            if (pc >= prog->len || seen[pc]) continue;
            seen[pc] = 1;
            if (pc == a) {
                loops = true;
                break;
            }
            const sre_insn_t &in = prog->insns[pc];
            if (in.opcode == SRE_OP_MATCH) continue;
            if (in.opcode == SRE_OP_JMP) {
                stack.push_back(in.x);
            } else if (in.opcode == SRE_OP_SPLIT) {
                stack.push_back(in.x);
                stack.push_back(in.y);
            } else {
                stack.push_back(pc + 1);
            }
        }
        if (loops) {
            *why = "a look-ahead assertion inside a loop (the VM's generation tags decide what its splice lists)";
            return NULL;
        }
    }

    /* bit numbering: pc 1 (the ".*?" ANY thread) first, then program order; a
     * thread that can consume '\n' and whose closure differs with ^ true gets a
     * twin bit right behind its own */
    sre_nfa_t *n = new sre_nfa_t();
    std::vector<int> bit_of(prog->len, -1), twin_of(prog->len, -1);
    auto needs_twin = [&](uint32_t pc) {
        const sre_insn_t &in = prog->insns[pc];
        if (in.opcode == SRE_OP_MATCH || !b.consumes(in, '\n') || fol[0][pc] == fol[1][pc]) return false;
        for (unsigned c = 0; c < 256; c++) {
            if (c != '\n' && b.consumes(in, c)) return true;
        }
        return false;       /* consumes nothing but '\n': ^ is always true behind it */
    };
    auto assign = [&](uint32_t pc) {
        if (bit_of[pc] >= 0) return;
        bit_of[pc] = (int) n->bit_pc.size();
        n->bit_pc.push_back(pc);
        if (needs_twin(pc)) {
            twin_of[pc] = (int) n->bit_pc.size();
            n->bit_pc.push_back(pc);
        }
    };
    if (prog->len > 1 && prog->insns[1].opcode == SRE_OP_ANY) assign(1);
    for (uint32_t pc : listable) assign(pc);
    n->nassert = (uint32_t) asserts.size();
    n->assert_slice = 0;
    /* the kernel variants are compiled for 1, 2, 3, 4, 6 or 8 byte slices */
    auto round_slices = [](uint32_t ns) { return ns <= 4 ? (ns ? ns : 1u) : ns <= 6 ? 6u : 8u; };
    if (n->nassert) {
        /* the assertions' bits: the LAST byte of the mask the kernel variant works on, a
         * byte of their own (the variant knows at compile time where to find them) */
        const uint32_t body = (uint32_t) (n->bit_pc.size() + 7) / 8;
        if (body + 1 > 8) {
            *why = "more than 64 thread bits (threads, newline twins, the assertions' byte)";
            delete n;
            return NULL;
        }
        n->assert_slice = round_slices(body + 1) - 1;
        while (n->bit_pc.size() < 8 * (size_t) n->assert_slice) n->bit_pc.push_back(0xffffffffu);   /* unused bits */
        for (uint32_t pc : asserts) {
            bit_of[pc] = (int) n->bit_pc.size();
            n->bit_pc.push_back(pc);
        }
    }
    n->nbits = (uint32_t) n->bit_pc.size();
    if (n->nbits > SRE_NFA_MAX_BITS) {
        *why = "more than 64 thread bits (threads plus their newline twins)";
        delete n;
        return NULL;
    }
    n->nslices = round_slices((n->nbits + 7) / 8);

    auto mask_of = [&](const std::set<uint32_t> &pcs) {
        uint64_t m = 0;
        for (uint32_t pc : pcs) {
            m |= 1ull << bit_of[pc];
            if (twin_of[pc] >= 0) m |= 1ull << twin_of[pc];
        }
        return m;
    };

    n->any_bits = 0;
    if (bit_of.size() > 1 && bit_of[1] >= 0) {
        n->any_bits = 1ull << bit_of[1];
        if (twin_of[1] >= 0) n->any_bits |= 1ull << twin_of[1];
    }
    n->match_bits = 0;
    memset(n->accept, 0, sizeof(n->accept));
    std::vector<uint64_t> fbit(n->nbits, 0);        /* follow mask per bit */
    for (uint32_t pc : listable) {
        const sre_insn_t &in = prog->insns[pc];
        if (in.opcode == SRE_OP_MATCH) {
            n->match_bits |= 1ull << bit_of[pc];
            continue;
        }
        for (unsigned c = 0; c < 256; c++) {
            if (!b.consumes(in, c)) continue;
            if (twin_of[pc] >= 0 && c == '\n') n->accept[c] |= 1ull << twin_of[pc];
            else n->accept[c] |= 1ull << bit_of[pc];
        }
        /* a thread that consumes nothing but '\n' always sees ^ true behind it */
        bool only_nl = b.consumes(in, '\n');
        for (unsigned c = 0; only_nl && c < 256; c++) {
            if (c != '\n' && b.consumes(in, c)) only_nl = false;
        }
        fbit[bit_of[pc]] = mask_of(fol[only_nl ? 1 : 0][pc]);
        if (twin_of[pc] >= 0) fbit[twin_of[pc]] = mask_of(fol[1][pc]);
    }

    for (int v = 0; v < 3; v++) {
        std::set<uint32_t> s;
        b.closure(0, v == 0, v != 2, s);
        n->init[v] = mask_of(s);
    }
    if (n->init[0] & n->match_bits) {
        *why = "nullable regex: the first match event is at offset 0, nothing to skip";
        delete n;
        return NULL;
    }

    /* ---- look-ahead assertions: byte kinds and the expansion tables */
    for (unsigned c = 0; c < 256; c++) {
        uint8_t k = sre_isword(c) ? SRE_NFA_KIND_WORD : c == '\n' ? SRE_NFA_KIND_NL : SRE_NFA_KIND_OTHER;
        bool    lead = false;
        if (prog->leading_byte != -1) lead = (int) c == prog->leading_byte;
        for (uint32_t i = 0; !lead && prog->leading_byte == -1 && i < prog->nleading; i++) {
            lead = b.consumes(prog->insns[prog->leading_insns[i]], c);
        }
        n->kind[c] = (uint8_t) (k | (lead ? SRE_NFA_LEADING : 0u));
    }
    if (n->nassert) {
        n->expand.assign(16 * 256, 0);
        for (uint32_t prev = 0; prev < 4; prev++) {
            for (uint32_t cur = 0; cur < 4; cur++) {
                const bool prev_word = prev == SRE_NFA_KIND_WORD, cur_word = cur == SRE_NFA_KIND_WORD;
                const bool at_start = prev == SRE_NFA_KIND_EDGE, at_end = cur == SRE_NFA_KIND_EDGE;
                auto holds = [&](uint8_t ch) {
                    switch (ch) {                                   /* :450-497 */
                    case SRE_ASSERT_SMALL_Z: return at_end;
                    case SRE_ASSERT_DOLLAR:  return at_end || cur == SRE_NFA_KIND_NL;
                    case SRE_ASSERT_SMALL_B: return prev_word != cur_word;
                    case SRE_ASSERT_BIG_B:   return prev_word == cur_word;
                    default:                 return false;
                    }
                };
                /* per assertion: everything its continuation lists at this position, and
                 * transitively what the assertions among THAT list (they see the same two
                 * bytes) */
                std::vector<uint64_t> xbit(n->nassert, 0);
                for (uint32_t i = 0; i < n->nassert; i++) {
                    if (!holds(prog->insns[asserts[i]].ch)) continue;
                    std::set<uint32_t> acc, todo;
                    todo.insert(asserts[i]);
                    std::set<uint32_t> done;
                    while (!todo.empty()) {
                        const uint32_t a = *todo.begin();
                        todo.erase(todo.begin());
                        if (!done.insert(a).second) continue;
                        std::set<uint32_t> cl;
                        b.closure(a + 1, at_start, at_start || prev == SRE_NFA_KIND_NL, cl);     /* :506-526 */
                        for (uint32_t pc : cl) {
                            acc.insert(pc);
                            const sre_insn_t &in = prog->insns[pc];
                            if (in.opcode == SRE_OP_ASSERT && holds(in.ch)) todo.insert(pc);
                        }
                    }
                    xbit[i] = mask_of(acc);
                }
                for (uint32_t v = 0; v < 256; v++) {
                    uint64_t m = 0;
                    for (uint32_t i = 0; i < n->nassert; i++) {
                        if ((v >> i) & 1) m |= xbit[i];
                    }
                    n->expand[(size_t) (prev * 4 + cur) * 256 + v] = m;
                }
            }
        }
    }

    n->follow.assign((size_t) n->nslices * 256, 0);
    for (uint32_t k = 0; k < n->nslices; k++) {
        for (uint32_t v = 0; v < 256; v++) {
            uint64_t m = 0;
            for (uint32_t j = 0; j < 8 && 8 * k + j < n->nbits; j++) {
                if ((v >> j) & 1) m |= fbit[8 * k + j];
            }
            n->follow[(size_t) k * 256 + v] = m;
        }
    }
    return n;
}
