#!/usr/bin/env python3
"""Instruction mix of the steady loops of one stream_kernel instantiation in a hipcc -S listing.
usage: isa_loops.py <file.s> <substring of the demangled kernel name> [--dump]"""
import collections
import re
import subprocess
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_util import innermost_loops  # noqa: E402


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
        for label, ins in innermost_loops(body):
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
            print("  loop %s: %d instrs, vmcnt waits %s: %s" % (label, len(ins), sorted(set(waits)), dict(ops)))
            if dump:
                print("\n".join("      " + i for i in ins))


if __name__ == "__main__":
    main()
