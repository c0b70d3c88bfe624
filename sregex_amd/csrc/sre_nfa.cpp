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
#include <stdlib.h>
#include <string.h>
#include <algorithm>
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
    if (nfa) delete nfa->sa;
    delete nfa;
}

static void build_shift_and(sre_nfa_t *n, const std::vector<uint64_t> &fbit, unsigned options);

extern "C" sre_nfa_t *
sre_nfa_build(const sre_program_t *prog, const char **why)
{
    const char *e = getenv("SRE_HIP_NFA_SA");
    return sre_nfa_build2(prog, e ? (unsigned) strtoul(e, NULL, 0) : 0u, why);
}

extern "C" sre_nfa_t *
sre_nfa_build2(const sre_program_t *prog, unsigned sa_options, const char **why)
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
    if (!(sa_options & SRE_NFA_SA_OFF)) build_shift_and(n, fbit, sa_options);
    return n;
}

/* ===================================================================== shift-and form */

namespace {

struct SaNode {
    uint64_t fol;           /* successors, as a mask over node ids (self excluded) */
    bool     self;          /* lists itself */
    bool     to_match;      /* its closure reaches MATCH */
    bool     is_match;      /* a sticky MATCH bit (accepts every byte, lists itself) */
    bool     is_any;        /* the explicit ".*?" thread (or its newline twin) */
    bool     is_assert;     /* a look-ahead assertion: consumes nothing, waits for the expansion */
    uint64_t acc[4];        /* bytes it consumes */
    int      next, prev;    /* the node one bit above / below (a link the shift serves) */
    int      gbit;          /* a bit of the plain form it stands for (-1: a MATCH node) */
};

struct SaLayout {
    bool                 ok;
    uint32_t             w64, carry, nbits, nlut, cost;
    uint32_t             hot[4];
    std::vector<int>     pos;       /* node -> bit */
};

inline int popc(uint64_t v) { return __builtin_popcountll(v); }

/* instructions per input byte of the device step (sre_hip_nfa.hip), the figure the options are compared by */
uint32_t
sa_cost(bool w64, bool carry, bool masked, bool evacc, uint32_t nlut)
{
    const uint32_t w = w64 ? 2 : 1;
    return 2 /* accept: address + read */ + 3 * w /* t, u, shift */ + (masked ? w : 0) + (evacc ? w : 0)
           + (carry ? 1 : 0) + (nlut ? 1 + 2 * nlut + (nlut - 1) * w : 0);
}

/* Place the chains: every chain is a run of consecutive bits, bottom to top; without `masked` one
 * hole above every chain (its top thread must not shift into a neighbour) unless it ends at the top
 * of a word.  The order is searched for the fewest bytes that hold a source. */
SaLayout
sa_place(const std::vector<std::vector<int>> &chains, const std::vector<uint8_t> &is_src, size_t nnodes,
         bool masked, bool force_w64, bool force_carry, int assert_chain)
{
    SaLayout best;
    best.ok = false;
    best.cost = ~0u;
    const size_t nc = chains.size();
    std::vector<size_t> order(nc);
    for (size_t i = 0; i < nc; i++) order[i] = i;
    /* chains with sources first, the ones with most sources per bit in front */
    std::vector<int> nsrc(nc, 0);
    for (size_t i = 0; i < nc; i++) {
        for (int v : chains[i]) nsrc[i] += is_src[v];
    }
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if ((nsrc[a] > 0) != (nsrc[b] > 0)) return nsrc[a] > 0;
        return (uint64_t) nsrc[a] * chains[b].size() > (uint64_t) nsrc[b] * chains[a].size();
    });
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto     rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int attempt = 0; attempt < 400; attempt++) {
        if (attempt) {
            /* perturb: a few random swaps of the best-known order early on, random shuffles later */
            const int swaps = attempt < 200 ? 1 + (int) (rnd() % 3) : (int) nc;
            for (int k = 0; k < swaps && nc > 1; k++) std::swap(order[rnd() % nc], order[rnd() % nc]);
        }
        if (assert_chain >= 0) {
            /* the look-ahead assertions: bits 0 .. of the mask (the kernel indexes the expansion table
             * with them: S & mask) */
            for (size_t k = 0; k < nc; k++) {
                if ((int) order[k] == assert_chain) {
                    std::swap(order[0], order[k]);
                    break;
                }
            }
        }
        for (int mode = 0; mode < 3; mode++) {
            /* mode 0: 32 bits; 1: two halves that shift alone; 2: 64 bits with a carry */
            if (mode == 0 && force_w64) continue;
            if (mode != 2 && force_carry) continue;
            std::vector<int> pos(nnodes, -1);
            uint32_t         at = 0;
            bool             fits = true;
            for (size_t ci = 0; ci < nc && fits; ci++) {
                const auto    &c = chains[order[ci]];
                const uint32_t len = (uint32_t) c.size();
                /* the look-ahead assertions share one byte of the mask */
                if ((int) order[ci] == assert_chain && (at & 7u) + len > 8) at = (at + 7u) & ~7u;
                if (mode == 1 && at < 32 && at + len > 32) at = 32;     /* a chain does not straddle the halves */
                const uint32_t limit = mode == 0 ? 32u : 64u;
                if (at + len > limit) {
                    fits = false;
                    break;
                }
                for (uint32_t j = 0; j < len; j++) pos[c[j]] = (int) (at + j);
                at += len;
                if (!masked && (int) order[ci] != assert_chain) {
                    /* the hole: not needed when the chain ends at the top of a word that drops the shifted-out bit */
                    const bool at_top = at == limit || (mode == 1 && at == 32);
                    if (!at_top) at++;
                }
            }
            if (!fits) continue;
            uint32_t hot_mask = 0;
            for (size_t v = 0; v < nnodes; v++) {
                if (is_src[v] && pos[v] >= 0) hot_mask |= 1u << (pos[v] >> 3);
            }
            const uint32_t nlut = (uint32_t) __builtin_popcount(hot_mask);
            if (nlut > SRE_NFA_SA_MAX_LUT) continue;
            const uint32_t cost = sa_cost(mode != 0, mode == 2, masked, false, nlut);
            uint32_t       hi = 0;
            for (size_t v = 0; v < nnodes; v++) hi = pos[v] + 1 > (int) hi ? (uint32_t) (pos[v] + 1) : hi;
            if (cost < best.cost) {
                best.ok = true;
                best.cost = cost;
                best.w64 = mode != 0;
                best.carry = mode == 2;
                best.nbits = hi;
                best.nlut = nlut;
                uint32_t k = 0;
                for (uint32_t b = 0; b < 8; b++) {
                    if (hot_mask & (1u << b)) best.hot[k++] = b;
                }
                best.pos = pos;
            }
            if (mode == 0) break;       /* fits 32 bits: the wider forms cannot be cheaper */
        }
        if (best.ok && best.nlut == 0 && !best.w64) break;
    }
    return best;
}

}  // namespace

static void
build_shift_and(sre_nfa_t *n, const std::vector<uint64_t> &fbit, unsigned options)
{
    const uint32_t nb = n->nbits;
    /* ---- the ".*?" thread stays implicit when it is one bit, always listed, and lists itself */
    bool implicit_any = popc(n->any_bits) == 1 && !(options & SRE_NFA_SA_EXPLICIT_ANY);
    int  any_bit = n->any_bits ? __builtin_ctzll(n->any_bits) : -1;
    if (implicit_any) {
        for (int v = 0; v < 3; v++) implicit_any = implicit_any && (n->init[v] & n->any_bits);
        implicit_any = implicit_any && (fbit[any_bit] & n->any_bits) && !(fbit[any_bit] & n->match_bits);
        for (unsigned c = 0; c < 256; c++) implicit_any = implicit_any && (n->accept[c] & n->any_bits);
    }
    const uint64_t drop = n->match_bits | (implicit_any ? n->any_bits : 0);

    /* ---- classes of equivalent threads: same follow set, same MATCH reach, listed by the same sets */
    std::vector<int> rep(nb);
    for (uint32_t i = 0; i < nb; i++) rep[i] = (int) i;
    std::vector<uint64_t> F(nb), A(nb * 4, 0);
    for (uint32_t i = 0; i < nb; i++) F[i] = fbit[i];
    for (unsigned c = 0; c < 256; c++) {
        for (uint32_t i = 0; i < nb; i++) {
            if ((n->accept[c] >> i) & 1) A[i * 4 + (c >> 6)] |= 1ull << (c & 63);
        }
    }
    uint64_t live = 0;
    for (uint32_t i = 0; i < nb; i++) {
        if (n->bit_pc[i] != 0xffffffffu && !((drop >> i) & 1)) live |= 1ull << i;
    }
    /* look-ahead assertions: the bits of the plain form's assertion byte */
    uint64_t assert_bits = 0;
    for (uint32_t j = 0; j < n->nassert; j++) assert_bits |= 1ull << (8 * n->assert_slice + j);
    if (n->nassert) options |= SRE_NFA_SA_FORCE_MASKED | SRE_NFA_SA_NO_EVACC;
    /* every set that lists threads: the initial lists, the ".*?" thread's closure, and what each
     * look-ahead assertion that holds puts into the list (per context) */
    std::vector<uint64_t> roots = {n->init[0], n->init[1], n->init[2], implicit_any ? fbit[any_bit] : 0};
    for (uint32_t ctx = 0; ctx < 16 && n->nassert; ctx++) {
        for (uint32_t j = 0; j < n->nassert; j++) roots.push_back(n->expand[(size_t) ctx * 256 + (1u << j)]);
    }
    bool     changed = !(options & SRE_NFA_SA_NO_MERGE);
    while (changed) {
        changed = false;
        for (uint32_t i = 0; i < nb && !changed; i++) {
            if (!((live >> i) & 1) || (((n->any_bits | assert_bits) >> i) & 1)) continue;
            for (uint32_t j = i + 1; j < nb && !changed; j++) {
                if (!((live >> j) & 1) || (((n->any_bits | assert_bits) >> j) & 1)) continue;
                if ((F[i] & ~n->match_bits) != (F[j] & ~n->match_bits)) continue;
                if (((F[i] & n->match_bits) != 0) != ((F[j] & n->match_bits) != 0)) continue;
                bool same = true;
                for (uint32_t k = 0; k < nb && same; k++) {
                    if (n->bit_pc[k] == 0xffffffffu) continue;
                    same = ((F[k] >> i) & 1) == ((F[k] >> j) & 1);
                }
                for (size_t r = 0; r < roots.size() && same; r++) same = ((roots[r] >> i) & 1) == ((roots[r] >> j) & 1);
                if (!same) continue;
                /* j joins i */
                for (uint32_t k = 0; k < nb; k++) {
                    if (rep[k] == (int) j) rep[k] = (int) i;
                    F[k] &= ~(1ull << j);
                }
                for (size_t r = 0; r < roots.size(); r++) roots[r] &= ~(1ull << j);
                for (int q = 0; q < 4; q++) A[i * 4 + q] |= A[j * 4 + q];
                live &= ~(1ull << j);
                changed = true;
            }
        }
    }

    /* ---- nodes */
    std::vector<SaNode> nodes;
    std::vector<int>    node_of(nb, -1);
    for (uint32_t i = 0; i < nb; i++) {
        if (!((live >> i) & 1)) continue;
        SaNode x;
        memset(&x, 0, sizeof(x));
        x.gbit = (int) i;
        x.next = x.prev = -1;
        x.is_any = (n->any_bits >> i) & 1;
        x.is_assert = (assert_bits >> i) & 1;
        x.to_match = (F[i] & n->match_bits) != 0;
        for (int q = 0; q < 4; q++) x.acc[q] = A[i * 4 + q];
        node_of[i] = (int) nodes.size();
        nodes.push_back(x);
    }
    if (nodes.size() > 64) return;
    auto to_nodes = [&](uint64_t gmask) {
        uint64_t m = 0;
        for (uint32_t i = 0; i < nb; i++) {
            if (((gmask >> i) & 1) && node_of[i] >= 0) m |= 1ull << node_of[i];
        }
        return m;
    };
    for (size_t v = 0; v < nodes.size(); v++) {
        const uint64_t f = to_nodes(F[nodes[v].gbit] & live);
        nodes[v].self = (f >> v) & 1;
        nodes[v].fol = f & ~(1ull << v);
    }
    const size_t nthreads = nodes.size();

    sre_nfa_sa_t *bestsa = NULL;
    for (int evacc = 0; evacc < 2; evacc++) {
        if (evacc == 0 && (options & SRE_NFA_SA_FORCE_EVACC)) continue;
        if (evacc == 1 && (options & SRE_NFA_SA_NO_EVACC)) continue;
        for (int masked = 0; masked < 2; masked++) {
            if (masked == 0 && (options & SRE_NFA_SA_FORCE_MASKED)) continue;
            std::vector<SaNode> nd(nodes.begin(), nodes.begin() + (long) nthreads);
            int shared_match = -1;
            if (!evacc) {
                /* MATCH as sticky bits: a private one above every thread that lists MATCH and nothing
                 * else, one shared bit (a lookup target) for the others */
                for (size_t v = 0; v < nthreads; v++) {
                    if (!nd[v].to_match) continue;
                    if (nd.size() >= 64) {
                        nd.resize(65);          /* no room for the MATCH bits */
                        break;
                    }
                    SaNode m;
                    memset(&m, 0, sizeof(m));
                    m.gbit = -1;
                    m.next = m.prev = -1;
                    m.is_match = true;
                    m.self = true;
                    for (int q = 0; q < 4; q++) m.acc[q] = ~0ull;
                    if (nd[v].fol == 0) {
                        m.prev = (int) v;
                        nd[v].next = (int) nd.size();
                        nd[v].fol |= 1ull << nd.size();
                        nd.push_back(m);
                    } else {
                        if (shared_match < 0) {
                            shared_match = (int) nd.size();
                            nd.push_back(m);
                        }
                        nd[v].fol |= 1ull << shared_match;
                    }
                }
                if (nd.size() > 64) continue;
                /* an expansion that lists MATCH is an event: it needs a bit to list */
                bool exp_match = false;
                for (size_t q = 0; q < n->expand.size() && !exp_match; q++) exp_match = (n->expand[q] & n->match_bits) != 0;
                if (exp_match && shared_match < 0) {
                    if (nd.size() >= 64) continue;
                    SaNode m;
                    memset(&m, 0, sizeof(m));
                    m.gbit = -1;
                    m.next = m.prev = -1;
                    m.is_match = true;
                    m.self = true;
                    for (int q = 0; q < 4; q++) m.acc[q] = ~0ull;
                    shared_match = (int) nd.size();
                    nd.push_back(m);
                }
            }
            const size_t nn = nd.size();
            /* links: a thread with ONE successor first (it then needs no lookup at all) */
            auto reaches = [&](int from, int target) {
                for (int k = from; k >= 0; k = nd[k].next) {
                    if (k == target) return true;
                }
                return false;
            };
            for (int pass = 0; pass < 2; pass++) {
                for (size_t v = 0; v < nn; v++) {
                    if (nd[v].next >= 0 || nd[v].fol == 0) continue;
                    if (pass == 0 && popc(nd[v].fol) != 1) continue;
                    for (size_t w = 0; w < nn; w++) {
                        if (!((nd[v].fol >> w) & 1) || nd[w].prev >= 0 || nd[w].is_assert || reaches((int) w, (int) v)) continue;
                        nd[v].next = (int) w;
                        nd[w].prev = (int) v;
                        break;
                    }
                }
            }
            std::vector<uint8_t> is_src(nn, 0);
            for (size_t v = 0; v < nn; v++) {
                uint64_t rest = nd[v].fol;
                if (nd[v].next >= 0) rest &= ~(1ull << nd[v].next);
                is_src[v] = rest != 0;
            }
            std::vector<std::vector<int>> chains;
            std::vector<int>              agroup;
            for (size_t v = 0; v < nn; v++) {
                if (nd[v].prev >= 0) continue;
                if (nd[v].is_assert) {
                    agroup.push_back((int) v);      /* (never linked: one group, one byte of the mask) */
                    continue;
                }
                std::vector<int> c;
                for (int k = (int) v; k >= 0; k = nd[k].next) c.push_back(k);
                chains.push_back(c);
            }
            int assert_chain = -1;
            if (!agroup.empty()) {
                assert_chain = (int) chains.size();
                chains.push_back(agroup);
            }
            SaLayout L = sa_place(chains, is_src, nn, masked != 0, (options & SRE_NFA_SA_FORCE_W64) != 0,
                                  (options & SRE_NFA_SA_FORCE_CARRY) != 0, assert_chain);
            if (!L.ok) continue;
            const uint32_t cost = sa_cost(L.w64, L.carry, masked, evacc, L.nlut);
            if (bestsa && bestsa->cost <= cost) continue;

            sre_nfa_sa_t *sa = new sre_nfa_sa_t();
            sa->cost = cost;
            sa->nbits = L.nbits;
            sa->w64 = L.w64;
            sa->carry = L.carry;
            sa->masked = (uint32_t) masked;
            sa->evacc = (uint32_t) evacc;
            sa->nlut = L.nlut;
            for (int k = 0; k < 4; k++) sa->hot[k] = k < (int) L.nlut ? L.hot[k] : 0;
            auto bits_of = [&](uint64_t node_mask) {
                uint64_t m = 0;
                for (size_t v = 0; v < nn; v++) {
                    if ((node_mask >> v) & 1) m |= 1ull << L.pos[v];
                }
                return m;
            };
            for (int v = 0; v < 3; v++) sa->init[v] = bits_of(to_nodes(n->init[v] & live));
            sa->seed = implicit_any ? bits_of(to_nodes(fbit[any_bit] & live)) : 0;
            memset(sa->accept, 0, sizeof(sa->accept));
            for (size_t v = 0; v < nn; v++) {
                const uint64_t b = 1ull << L.pos[v];
                sa->valid |= b;
                if (nd[v].self) sa->self |= b;
                if (nd[v].next >= 0) sa->shift_src |= b;
                if (nd[v].is_any) sa->any_bits |= b;
                if (nd[v].is_match) sa->match_bits |= b;
                if (evacc && nd[v].to_match) sa->msrc |= b;
                for (unsigned c = 0; c < 256; c++) {
                    if ((nd[v].acc[c >> 6] >> (c & 63)) & 1) sa->accept[c] |= b;
                }
            }
            sa->lut.assign((size_t) L.nlut * 256, 0);
            for (uint32_t k = 0; k < L.nlut; k++) {
                for (uint32_t x = 0; x < 256; x++) {
                    uint64_t m = 0;
                    for (size_t v = 0; v < nn; v++) {
                        if ((uint32_t) (L.pos[v] >> 3) != L.hot[k] || !((x >> (L.pos[v] & 7)) & 1)) continue;
                        uint64_t rest = nd[v].fol;
                        if (nd[v].next >= 0) rest &= ~(1ull << nd[v].next);
                        m |= bits_of(rest);
                    }
                    sa->lut[(size_t) k * 256 + x] = m;
                }
            }
            sa->nassert = n->nassert;
            sa->assert_byte = 0;
            if (n->nassert) {
                sa->assert_byte = (uint32_t) L.pos[agroup[0]] >> 3;
                sa->expand.assign(16 * 256, 0);
                for (uint32_t ctx = 0; ctx < 16; ctx++) {
                    std::vector<uint64_t> xsa(n->nassert, 0);
                    for (uint32_t j = 0; j < n->nassert; j++) {
                        const uint64_t g = n->expand[(size_t) ctx * 256 + (1u << j)];
                        xsa[j] = bits_of(to_nodes(g & live));
                        if ((g & n->match_bits) && shared_match >= 0) xsa[j] |= 1ull << L.pos[shared_match];
                    }
                    for (uint32_t x = 0; x < 256; x++) {
                        uint64_t m = 0;
                        for (uint32_t j = 0; j < n->nassert; j++) {
                            const int apos = L.pos[node_of[8 * n->assert_slice + j]];
                            if ((x >> (apos & 7)) & 1) m |= xsa[j];
                        }
                        sa->expand[(size_t) ctx * 256 + x] = m;
                    }
                }
            }
            sa->bit_of.assign(nb, -3);
            for (uint32_t i = 0; i < nb; i++) {
                if (n->bit_pc[i] == 0xffffffffu) continue;
                if ((n->match_bits >> i) & 1) sa->bit_of[i] = -2;
                else if (implicit_any && (int) i == any_bit) sa->bit_of[i] = -1;
                else sa->bit_of[i] = L.pos[node_of[rep[i]]];
            }
            delete bestsa;
            bestsa = sa;
        }
    }
    n->sa = bestsa;
}
