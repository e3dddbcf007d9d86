#!/usr/bin/env python3
"""Static opcode histogram of one kernel in a hipcc -save-temps .s file:  tools/isa_mix.py <file.s> <mangled-name-substring> [top]"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
starts = [i for i, l in enumerate(lines) if pat in l and l.rstrip().split(";")[0].strip().endswith(":") and not l.startswith(".")]
for st in starts:
    end = next(i for i in range(st, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    cnt = collections.Counter()
    for l in lines[st + 1:end]:
        l = l.strip()
        if not l or l[0] in ";." or l.split(";")[0].strip().endswith(":"):
            continue
        cnt[l.split()[0]] += 1
    print(lines[st].split(":")[0], sum(cnt.values()), "instructions")
    print("  ", ", ".join("%s %d" % kv for kv in cnt.most_common(top)))
    meta = [l.strip() for l in lines[end:end + 60] if "NumVgprs" in l or "ScratchSize" in l or "Occupancy" in l or "LDSByteSize" in l]
    print("  ", " | ".join(meta))
