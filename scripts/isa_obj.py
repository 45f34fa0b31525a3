"""Loops of a gfx950 CODE OBJECT (llvm-objdump -d), by control-flow graph: what tests/test_jit_isa.py runs over the kernels
the run-time compiler (hiprtc, rf_jit.cpp) produces, where no `hipcc -S` listing with loop comments exists.

functions(path) -> {mangled name: [Ins]};  loops(ins) -> [ [Ins in execution order of one iteration] ] for every loop that is
a SIMPLE CYCLE of basic blocks (each block has exactly one successor inside the loop: the branch-free steady loops of the
stream kernel are); other strongly connected regions come back as (None, blocks) through regions()."""
import collections
import re
import subprocess

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
Ins = collections.namedtuple("Ins", "addr op text target")


def functions(path):
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
    funcs, name, body = {}, None, []
    for line in out.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
        if m:
            name, body = m.group(2), []
            funcs[name] = body
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+): [0-9A-F ]+(?:<(\S+?)(?:\+0x([0-9a-f]+))?>)?\s*$", line)
        if m and name is not None:
            op, rest, addr, sym, off = m.groups()
            target = None
            if op.startswith(("s_cbranch", "s_branch")) and sym is not None:
                target = (sym, int(off, 16) if off else 0)
            body.append(Ins(int(addr, 16), op, (op + " " + rest).strip(), target))
    # branch targets are function-relative offsets: make them absolute
    res = {}
    for fn, body in funcs.items():
        if not body:
            continue
        base = body[0].addr
        res[fn] = [i._replace(target=(base + i.target[1]) if (i.target and i.target[0] == fn) else None) if i.target else i for i in body]
    return res


def blocks(ins):
    """basic blocks: {start addr: ([Ins], [successor start addrs])}"""
    starts = {ins[0].addr}
    for k, i in enumerate(ins):
        if i.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            if k + 1 < len(ins):
                starts.add(ins[k + 1].addr)
            if i.target is not None:
                starts.add(i.target)
    out, cur = {}, None
    for i in ins:
        if i.addr in starts:
            cur = i.addr
            out[cur] = ([], [])
        out[cur][0].append(i)
    order = sorted(out)
    for n, a in enumerate(order):
        body, succ = out[a]
        last = body[-1]
        nxt = order[n + 1] if n + 1 < len(order) else None
        if last.op.startswith("s_branch"):
            if last.target is not None:
                succ.append(last.target)
        elif last.op.startswith("s_cbranch"):
            if last.target is not None:
                succ.append(last.target)
            if nxt is not None:
                succ.append(nxt)
        elif last.op.startswith(("s_endpgm", "s_setpc")):
            pass
        elif nxt is not None:
            succ.append(nxt)
    return out


def _sccs(nodes, succ):
    """strongly connected components (iterative Tarjan) of the subgraph `nodes` with successor lists succ[a] (restricted to nodes)"""
    index, low, onstack, stack, res, counter = {}, {}, set(), [], [], [0]
    for root in nodes:
        if root in index:
            continue
        work = [(root, iter(succ[root]))]
        index[root] = low[root] = counter[0]; counter[0] += 1
        stack.append(root); onstack.add(root)
        while work:
            v, it = work[-1]
            advanced = False
            for w in it:
                if w not in nodes:
                    continue
                if w not in index:
                    index[w] = low[w] = counter[0]; counter[0] += 1
                    stack.append(w); onstack.add(w)
                    work.append((w, iter(succ[w])))
                    advanced = True
                    break
                if w in onstack:
                    low[v] = min(low[v], index[w])
            if advanced:
                continue
            work.pop()
            if work:
                low[work[-1][0]] = min(low[work[-1][0]], low[v])
            if low[v] == index[v]:
                comp = set()
                while True:
                    w = stack.pop(); onstack.discard(w); comp.add(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in succ[v]:
                    res.append(comp)
    return res


def regions(ins):
    """(blocks, INNERMOST cyclic regions): a strongly connected region whose cycles all pass through its entry blocks; outer loops
    are peeled by cutting the edges back into their entries and looking again (the stream kernel's walks sit inside one outer
    loop over work items)"""
    bl = blocks(ins)
    succ = {a: [t for t in bl[a][1] if t in bl] for a in bl}
    out = []

    def walk(nodes, succ):
        for comp in _sccs(nodes, succ):
            entries = {a for a in comp if any(a in succ[p] for p in nodes if p not in comp)} or {min(comp)}
            cut = {a: [t for t in succ[a] if not (t in entries and a in comp)] for a in comp}
            inner = [c for c in _sccs(comp, cut)]
            if not inner:
                out.append(comp)
            else:
                walk(comp, cut)

    walk(set(bl), succ)
    return bl, out


def loops(ins):
    """[(is_simple_cycle, [Ins] in execution order from the loop's entry block)] for every cyclic region"""
    bl, regs = regions(ins)
    out = []
    for comp in regs:
        inner = {a: [s for s in bl[a][1] if s in comp] for a in comp}
        entries = sorted(a for a in comp if any(a in bl[p][1] for p in bl if p not in comp))
        simple = all(len(v) == 1 for v in inner.values())
        if simple and entries:
            seq, a = [], entries[0]
            for _ in range(len(comp)):
                seq += bl[a][0]
                a = inner[a][0]
            out.append((True, seq))
        else:
            # a loop with inner branches (the exec-masked store of the stream kernel): the entry block first, then the blocks
            # laid out after it, then the ones before it -- the order tests/test_isa_invariants.py uses on hipcc -S listings
            order = sorted(comp)
            if entries:
                k = order.index(entries[0])
                order = order[k:] + order[:k]
            out.append((False, [i for a in order for i in bl[a][0]]))
    return out
