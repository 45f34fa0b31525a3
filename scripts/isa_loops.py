#!/usr/bin/env python3
"""Instruction mix of the steady loops of one stream_kernel instantiation in a hipcc -S listing.
usage: isa_loops.py <file.s> <substring of the demangled kernel name> [--dump]"""
import collections
import re
import subprocess
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    dump = "--dump" in sys.argv
    name, body, kernels = None, [], {}
    for l in lines:
        m = re.match(r"^(_ZN2rf13stream_kernel\w+):", l)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(l)
            if "s_endpgm" in l:
                kernels[name] = body
                name = None
    dem = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.split("\n")
    for (mangled, body), d in zip(kernels.items(), dem):
        d = re.sub(r"rf::|void |\(rf::StreamArgs<.*", "", d)
        if want not in d:
            continue
        print("==", d)
        spans = []
        for h, l in enumerate(body):
            if not re.match(r"^\.LBB\d+_\d+:.*Loop Header", l):
                continue
            label = l.split(":")[0]
            back = [k for k in range(h, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[k])]
            if back:
                spans.append((h, back[-1]))
        for h, e in spans:
            ins = [x.strip() for x in body[h + 1:e + 1] if x.strip() and not x.strip().startswith((";", "."))]
            text = "\n".join(ins)
            waits = re.findall(r"vmcnt\((\d+)\)", text)
            if not waits or "global_store" not in text:
                continue
            ops = collections.Counter()
            for i in ins:
                op = i.split()[0]
                cls = ("pk_fma" if op == "v_pk_fma_f32" else "v_mov" if op.startswith("v_mov") or op.startswith("v_accvgpr") else "valu" if op.startswith("v_") else
                       "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write") else "salu" if op.startswith("s_") else op)
                ops[cls] += 1
            print("  loop @%d: %d instrs, vmcnt waits %s: %s" % (h, len(ins), sorted(set(waits)), dict(ops)))
            if dump:
                print("\n".join("      " + i for i in ins))


if __name__ == "__main__":
    main()
