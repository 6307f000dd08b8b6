#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file (static counts, per basic block).
usage: isa_stats.py file.s kernel_substring [--blocks]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[\w$.]+:", l) and key in l and not l.startswith("."))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].strip().startswith("s_endpgm") and False) if False else None
    tot = collections.Counter()
    blocks = []
    cur_name, cur = "entry", collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".section") or t.startswith(".rodata") or t.startswith(".amdhsa_kernel"):
            break
        m = re.match(r"^(\.LBB[\w]+):", t)
        if m:
            blocks.append((cur_name, cur))
            cur_name, cur = m.group(1), collections.Counter()
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k = "vmem"
        else:
            k = "other"
        cur[k] += 1
        tot[k] += 1
        tot["op:" + op] += 1
    blocks.append((cur_name, cur))
    print("total", {k: v for k, v in tot.items() if not k.startswith("op:")})
    if show_blocks:
        for n, c in blocks:
            if sum(c.values()) >= 8:
                print("%-12s %s" % (n, dict(c)))
    top = sorted(((v, k[3:]) for k, v in tot.items() if k.startswith("op:")), reverse=True)[:40]
    print(" ".join("%s:%d" % (k, v) for v, k in top))


main()
