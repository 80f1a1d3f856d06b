#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps .s listing (static counts; loops counted once).
usage: isa_hist.py file.s <substring of the mangled kernel name> [--top N]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l]
    for i, name in starts:
        hist = collections.Counter()
        for l in lines[i + 1:]:
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                break
            m = re.match(r"^\t([a-z_0-9]+)", l)
            if m and not m.group(1).startswith("."):
                hist[m.group(1)] += 1
        tot = sum(hist.values())
        cls = collections.Counter()
        for k, v in hist.items():
            if k.startswith("v_") and "f64" in k:
                cls["valu_f64"] += v
            elif k.startswith("v_"):
                cls["valu_other"] += v
            elif k.startswith("ds_"):
                cls["lds"] += v
            elif k.startswith("s_"):
                cls["salu"] += v
            elif k.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cls["vmem"] += v
            else:
                cls["other"] += v
        print(name, "total", tot, dict(cls))
        for k, v in hist.most_common(top):
            print("   %-28s %d" % (k, v))


main()
