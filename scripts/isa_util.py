"""Loops of a hipcc -S listing by BASIC BLOCK membership (not by layout): hipcc marks every block of a loop with the
comment `in Loop: Header=BB<f>_<n>` and is free to place a loop's latch in front of its header, so "from the header
label to the last branch back" misses rotated loops.  Shared by scripts/isa_loops.py and tests/test_isa_invariants.py."""
import re


def innermost_loops(body):
    """[(header label, instruction lines in EXECUTION order of one iteration: header block first, then the blocks
    laid out after it, then the ones laid out before it)] for every innermost loop of one function body."""
    blocks = []                       # (label or None, comment, [lines])
    cur = (None, "", [])
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        m2 = re.match(r"^; %bb\.\d+:(.*)$", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), m.group(2), [])
        elif m2:
            blocks.append(cur)
            cur = (None, m2.group(1), [])
        elif re.match(r"^\s+; (=>|  )", l) and not cur[2]:
            # continuation of the block's header comment: a NESTED loop's header carries "Parent Loop ..." on the label line and
            # "=>  This Inner Loop Header: Depth=2" on the next
            cur = (cur[0], cur[1] + " " + l.strip(), cur[2])
        else:
            cur[2].append(l)
    blocks.append(cur)
    out = []
    for i, (label, comment, lines) in enumerate(blocks):
        if label is None or "Inner Loop Header" not in comment:
            continue
        tag = "Header=BB" + label[len(".LBB"):] + " "
        after = [b for b in blocks[i + 1:] if tag in b[1] + " "]
        before = [b for b in blocks[:i] if tag in b[1] + " "]
        seq = list(lines)
        for b in after + before:
            seq += b[2]
        out.append((label, [x.strip() for x in seq if x.strip() and not x.strip().startswith((";", "."))]))
    return out


def scalar_base_violations(instrs):
    """instrs: instruction texts of one kernel in layout order.  On gfx9 a vector-memory instruction that reads an SGPR needs five
    wait states behind a VECTOR instruction that wrote it (v_readlane reloading a spilled scalar, v_readfirstlane, v_cmp); hipcc
    cannot see the global_* instructions the kernels issue from asm statements, so every one of them must address through VCC,
    loaded by an s_mov_b64 a few instructions in front of it INSIDE the same asm (RF_SBASE, rf_device.h): a scalar read of the
    reloaded pair is interlocked, a scalar write followed by the memory instruction has no hazard.  Returns the instructions that
    do not."""
    bad = []
    for k, ins in enumerate(instrs):
        if not ins.startswith("global_"):
            continue
        ops = [o.strip() for o in ins.split(None, 1)[1].split(",")] if " " in ins else []
        base = ops[-1].split()[0] if ops else ""
        if base != "vcc" or not any(p.startswith("s_mov_b64 vcc, s[") for p in instrs[max(0, k - 6):k]):
            bad.append(ins)
    return bad
