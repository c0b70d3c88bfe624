/*
 * sre_pwave.cpp — builds the wave form of a program (sre_pwave.h): static closure lists.
 */
#include "sre_pwave.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

struct Walk {
    const sre_program_t *prog;
    std::vector<int>     tid_of;        /* pc -> thread id, -1 */
    bool                 a_ok, caret_ok, from_loop;
    /* results */
    std::vector<sre_pwave_entry_t> out;
    bool                 done = false, sss = false;
    sre_pwave_entry_t    match_entry;

    /* the reference's add_thread (sre_vm_pike.c:756-942) from pc0 with fresh generation tags, as
     * sre_hip_vm.hip Pike::closure() restates it: explicit stack, RESTORE records undo SAVEs */
    void run(uint32_t pc0)
    {
        struct Rec { uint32_t a; bool restore; bool old; };
        std::vector<Rec>     stack;
        std::vector<uint8_t> tag(prog->len + 1, 0);
        uint64_t             saved = 0;
        uint32_t             pc = pc0;
        bool                 restart = true;
        for (;;) {
            if (!restart) {
                for (;;) {
                    if (stack.empty()) return;
                    Rec r = stack.back();
                    stack.pop_back();
                    if (r.restore) {
                        saved = r.old ? (saved | (1ull << r.a)) : (saved & ~(1ull << r.a));
                    } else {
                        pc = r.a;
                        break;
                    }
                }
            }
            restart = false;
            for (;;) {
                if (pc >= prog->len) break;
                const sre_insn_t &in = prog->insns[pc];
                bool              list_it = false;
                if (tag[pc]) {
                    if (in.opcode == SRE_OP_SPLIT && !tag[in.y]) {      /* :774-784 */
                        if (pc == 0) sss = true;
                        pc = in.y;
                        continue;
                    }
                    break;
                }
                tag[pc] = 1;
                switch (in.opcode) {
                case SRE_OP_JMP:
                    pc = in.x;
                    continue;
                case SRE_OP_SPLIT:
                    if (pc == 0) sss = true;                            /* :799-802 */
                    stack.push_back(Rec{in.y, false, false});
                    pc = in.x;
                    continue;
                case SRE_OP_SAVE:
                    stack.push_back(Rec{in.arg, true, ((saved >> in.arg) & 1) != 0});
                    saved |= 1ull << in.arg;
                    pc = pc + 1;
                    continue;
                case SRE_OP_ASSERT:
                    if (in.ch == SRE_ASSERT_BIG_A) {
                        if (!a_ok) break;
                        pc = pc + 1;
                        continue;
                    }
                    if (in.ch == SRE_ASSERT_CARET) {
                        if (!caret_ok) break;
                        pc = pc + 1;
                        continue;
                    }
                    list_it = true;         /* (look-ahead: such programs have no wave form) */
                    break;
                case SRE_OP_MATCH:
                    if (from_loop) {
                        /* :895-898 SRE_DONE: the walk ends here */
                        done = true;
                        match_entry.tid = (uint16_t) tid_of[pc];
                        match_entry.saves = saved;
                        return;
                    }
                    list_it = true;
                    break;
                default:
                    list_it = true;
                    break;
                }
                if (list_it) {
                    sre_pwave_entry_t e;
                    memset(&e, 0, sizeof(e));
                    e.tid = (uint16_t) tid_of[pc];
                    e.saves = saved;
                    out.push_back(e);
                }
                break;
            }
        }
    }
};

bool
consumes(const sre_program_t *prog, const sre_insn_t &in, unsigned c)
{
    switch (in.opcode) {
    case SRE_OP_CHAR:  return c == in.ch;
    case SRE_OP_ANY:   return true;
    case SRE_OP_IN:    return sre_in_ranges(&prog->ranges[in.x], in.nranges, c) != 0;
    case SRE_OP_NOTIN: return sre_in_ranges(&prog->ranges[in.x], in.nranges, c) == 0;
    default:           return false;
    }
}

}  // namespace

extern "C" sre_pwave_hdr_t *
sre_pwave_build(const sre_program_t *prog)
{
    if (prog->lookahead_asserts || prog->nslots > SRE_PWAVE_MAX_SLOTS || prog->len > 4096) return NULL;
    std::vector<int>      tid_of(prog->len, -1);
    std::vector<uint32_t> tid_pc;
    for (uint32_t pc = 0; pc < prog->len; pc++) {
        switch (prog->insns[pc].opcode) {
        case SRE_OP_CHAR: case SRE_OP_IN: case SRE_OP_NOTIN: case SRE_OP_ANY: case SRE_OP_MATCH:
            tid_of[pc] = (int) tid_pc.size();
            tid_pc.push_back(pc);
            break;
        default:
            break;
        }
    }
    if (tid_pc.size() > SRE_PWAVE_MAX_THREADS) return NULL;
    const uint32_t nt = (uint32_t) tid_pc.size();

    std::vector<sre_pwave_list_t>  lists;
    std::vector<sre_pwave_entry_t> entries;
    auto add_lists = [&](uint32_t pc0, bool from_loop) {
        for (uint32_t ctx = 0; ctx < SRE_PWAVE_NCTX; ctx++) {
            Walk w;
            w.prog = prog;
            w.tid_of = tid_of;
            w.a_ok = ctx == 2;
            w.caret_ok = ctx >= 1;
            w.from_loop = from_loop;
            memset(&w.match_entry, 0, sizeof(w.match_entry));
            w.run(pc0);
            sre_pwave_list_t l;
            l.off = (uint32_t) entries.size();
            l.len = (uint16_t) w.out.size();
            l.done = w.done ? 1 : 0;
            l.sss = w.sss ? 1 : 0;
            entries.insert(entries.end(), w.out.begin(), w.out.end());
            if (w.done) entries.push_back(w.match_entry);
            lists.push_back(l);
        }
    };
    add_lists(0, false);
    std::vector<uint16_t> tid_list(nt, 0xffffu);
    uint32_t              nl = 1;
    for (uint32_t t = 0; t < nt; t++) {
        if (prog->insns[tid_pc[t]].opcode == SRE_OP_MATCH) continue;
        tid_list[t] = (uint16_t) nl++;
        add_lists(tid_pc[t] + 1, true);
    }

    const size_t off_lists = (sizeof(sre_pwave_hdr_t) + 15) & ~(size_t) 15;
    const size_t off_entries = (off_lists + lists.size() * sizeof(sre_pwave_list_t) + 15) & ~(size_t) 15;
    const size_t off_ncaps = off_entries + (entries.size() + 1) * sizeof(sre_pwave_entry_t);
    const size_t bytes = (off_ncaps + (size_t) prog->nregexes * 4 + 15) & ~(size_t) 15;
    sre_pwave_hdr_t *h = static_cast<sre_pwave_hdr_t *>(calloc(1, bytes));
    if (h == NULL) return NULL;
    h->nthreads = nt;
    h->nslots = prog->nslots;
    h->nlists = nl;
    h->nentries = (uint32_t) entries.size();
    h->nleading = prog->nleading;
    h->bytes = (uint32_t) bytes;
    h->off_lists = (uint32_t) off_lists;
    h->off_entries = (uint32_t) off_entries;
    h->multi_ncaps_off = (uint32_t) off_ncaps;
    h->nregexes = prog->nregexes;
    for (uint32_t t = 0; t < nt; t++) {
        const sre_insn_t &in = prog->insns[tid_pc[t]];
        h->tid_pc[t] = tid_pc[t];
        h->tid_list[t] = tid_list[t];
        h->tid_match[t] = in.opcode == SRE_OP_MATCH ? (uint16_t) (in.arg + 1) : 0;
        for (unsigned c = 0; c < 256; c++) {
            if (consumes(prog, in, c)) h->accept[t][c >> 5] |= 1u << (c & 31);
        }
    }
    for (unsigned c = 0; c < 256; c++) {
        bool lead = false;
        if (prog->leading_byte != -1) {
            lead = (int) c == prog->leading_byte;
        } else {
            for (uint32_t i = 0; !lead && i < prog->nleading; i++) lead = consumes(prog, prog->insns[prog->leading_insns[i]], c);
        }
        if (lead) h->lead[c >> 5] |= 1u << (c & 31);
    }
    uint8_t *base = reinterpret_cast<uint8_t *>(h);
    memcpy(base + off_lists, lists.data(), lists.size() * sizeof(sre_pwave_list_t));
    memcpy(base + off_entries, entries.data(), entries.size() * sizeof(sre_pwave_entry_t));
    memcpy(base + off_ncaps, prog->multi_ncaps, (size_t) prog->nregexes * 4);
    return h;
}
