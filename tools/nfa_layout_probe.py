"""Development probe: how would the compiled programs lay out as a shift-and NFA?
For every reference block (and a few hand-picked programs) parse the program dump, build
the set-level follow relation, merge equivalent threads and report bits / exceptions."""
import collections
import re
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sregex_amd as S
import harness


def parse_dump(text):
    insns = []
    for line in text.splitlines():
        m = re.match(r"\s*(\d+)\. (\w+)\s*(.*)$", line)
        if not m:
            continue
        op, arg = m.group(2), m.group(3)
        insns.append((op, arg))
    return insns


def accept_set(op, arg):
    if op == "char":
        return frozenset([int(arg)])
    if op == "any":
        return frozenset(range(256))
    if op in ("in", "notin"):
        s = set()
        for r in arg.split(","):
            a, b = r.strip().split("-")
            s.update(range(int(a), int(b) + 1))
        return frozenset(s if op == "in" else set(range(256)) - s)
    return None


def closure(insns, pc0):
    out, seen, stack = set(), set(), [pc0]
    while stack:
        pc = stack.pop()
        while pc < len(insns) and pc not in seen:
            seen.add(pc)
            op, arg = insns[pc]
            if op == "jmp":
                pc = int(arg)
            elif op == "split":
                x, y = [int(v) for v in arg.split(",")]
                stack.append(y)
                pc = x
            elif op == "save":
                pc += 1
            elif op == "assert":
                return None         # not handled by the probe
            else:
                out.add(pc)
                break
    return out


def analyse(insns):
    """-> dict or None (assertions)"""
    threads = [pc for pc, (op, _) in enumerate(insns) if op in ("char", "any", "in", "notin", "match")]
    fol = {}
    for pc in threads:
        if insns[pc][0] == "match":
            fol[pc] = frozenset(["M"])
            continue
        c = closure(insns, pc + 1)
        if c is None:
            return None
        fol[pc] = frozenset("M" if insns[q][0] == "match" else q for q in c)
    init = closure(insns, 0)
    if init is None:
        return None
    init = frozenset("M" if insns[q][0] == "match" else q for q in init)
    nodes = [pc for pc in threads if insns[pc][0] != "match"]
    anypc = 1 if len(insns) > 1 and insns[1][0] == "any" else None
    # the .*? thread is implicit (always alive): drop it; its follow is the seed
    seed = fol.get(anypc, init) if anypc is not None else init
    nodes = [pc for pc in nodes if pc != anypc]
    acc = {pc: accept_set(*insns[pc]) for pc in nodes}
    F = {pc: frozenset(x for x in fol[pc] if x != anypc) for pc in nodes}
    # merge: same follow (self-reference normalised), same membership everywhere
    changed = True
    rep = {pc: pc for pc in nodes}
    while changed:
        changed = False
        live = sorted(set(rep.values()))
        def norm(pc):
            return frozenset(("SELF" if rep.get(x, x) == pc else rep.get(x, x)) for x in F[pc])
        member = collections.defaultdict(set)
        for src in live:
            for x in F[src]:
                if x != "M":
                    member[rep[x]].add(src)
        for x in seed:
            if x != "M" and x != anypc:
                member[rep[x]].add("SEED")
        groups = collections.defaultdict(list)
        for pc in live:
            groups[(norm(pc), frozenset(member[pc] - {pc}), pc in member[pc])].append(pc)
        for g in groups.values():
            if len(g) > 1:
                for pc in g[1:]:
                    for k in list(rep):
                        if rep[k] == pc:
                            rep[k] = g[0]
                    acc[g[0]] = acc[g[0]] | acc[pc]
                changed = True
        if changed:
            F = {pc: frozenset(rep.get(x, x) for x in F[pc]) for pc in set(rep.values())}
    live = sorted(set(rep.values()))
    # layout: greedy chains.  thread i is plain when follow \ {self} is empty or {j} with j placed at i+1
    succ = {}
    for pc in live:
        rest = [x for x in F[pc] if x != pc]
        succ[pc] = rest
    # choose for every node with exactly one non-self successor an edge pc -> j; a j can be the
    # "next" of one pc only, and the edges must form paths (no cycles)
    nxt, prev = {}, {}
    for pc in live:
        if len(succ[pc]) == 1:
            j = succ[pc][0]
            if j in prev or j == pc:
                continue
            # cycle check
            k, cyc = j, False
            while k in nxt:
                k = nxt[k]
                if k == pc:
                    cyc = True
                    break
            if cyc or j == pc:
                continue
            nxt[pc] = j
            prev[j] = pc
    exc = [pc for pc in live if succ[pc] and pc not in nxt]
    # multi-successor nodes: could shift into ONE of their successors and leave the rest to the table;
    # still exceptions.  Fan-in exceptions (one successor, already taken):
    fanin = [pc for pc in exc if len(succ[pc]) == 1]
    nbits = len(live) + 1   # + one MATCH bit
    return dict(nthreads=len(threads), nbits=nbits, nexc=len(exc), nfanin=len(fanin),
                nself=sum(1 for pc in live if pc in F[pc]))


def main():
    rows = []
    zoo = [[rb"(?:a|b)*a(?:a|b){7}@"], [rb"(?:a|b)*a(?:a|b){27}@"], [rb"[a-c]{20,56}@"],
           [rb"a", rb"ab", rb"c", rb"a(bc)", rb"e(f)", rb"gh", rb"A", rb"b", rb"BLAH", rb"\s+", rb"abcd", rb"bc"],
           [rb"([a-z]+)://([^/ ]+)(/[^ ?]*)?(\?[^ ]*)?"], [rb"x.{30}y"], [rb"(?:GET|POST|PUT|HEAD) /[a-z0-9/]{1,40}\.(?:html|php|js|css) HTTP/1\.[01]"]]
    for pats in zoo:
        with S.Pool() as pool:
            prog = S.compile(pool, S.parse(pool, pats))
            r = analyse(parse_dump(prog.dump()))
            print(pats[0][:40], len(pats), r)
    hist = collections.Counter()
    big = []
    for blk in harness.load_blocks():
        for name, regexes, flags, multi, ref in harness.block_variants(blk):
            if ref["rc"] != 0:
                continue
            with S.Pool() as pool:
                try:
                    prog = S.compile(pool, S.parse(pool, regexes, flags, multi))
                except Exception:
                    continue
                r = analyse(parse_dump(prog.dump()))
                if r is None:
                    hist["assert"] += 1
                    continue
                if r["nthreads"] >= 24:
                    big.append((r["nthreads"], r["nbits"], r["nexc"], r["nfanin"], r["nself"], regexes[-1][:50]))
                hist[(min(r["nbits"] // 8, 9), min(r["nexc"] // 4, 9))] += 1
    print(sorted(hist.items(), key=str))
    for b in sorted(big)[-60:]:
        print(b)


if __name__ == "__main__":
    main()
