#!/usr/bin/env python3
"""VGPR / SGPR / scratch / occupancy / LDS per kernel from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (compile a .hip file with that flag and pass the stderr log).  usage: kernel_resources.py <log> [substring]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, d in zip(blocks, dem):
        g = lambda pat: int(re.search(pat, b).group(1))
        short = re.sub(r"rf::|void |\(rf::StreamArgs<.*", "", d)
        if want in short:
            print("v=%3d s=%3d scratch=%d occ=%d lds=%6d  %s" % (g(r"VGPRs: (\d+)"), g(r"SGPRs: (\d+)"), g(r"ScratchSize \[bytes/lane\]: (\d+)"),
                                                              g(r"Occupancy \[waves/SIMD\]: (\d+)"), g(r"LDS Size \[bytes/block\]: (\d+)"), short[:170]))


if __name__ == "__main__":
    main()
