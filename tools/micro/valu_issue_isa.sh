#!/bin/bash
# Disassemble build/valu_issue's gfx950 code object and print, per kernel, the timed loop body (between the two
# s_memtime stamps) with an opcode histogram: the evidence that the timed loops hold the instructions they are named for.
# usage: tools/micro/valu_issue_isa.sh > profiles/<tag>_valu_issue_isa.txt
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
TMP=$R/build/valu_issue_co
mkdir -p $TMP
cd $TMP
rm -f *.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$R/build/valu_issue >/dev/null 2>&1 || true
# the fat binary section holds one gfx950 code object: extract it with roc-obj tools if present, else via objcopy
if command -v roc-obj-ls >/dev/null 2>&1; then
  roc-obj -t gfx950 -o $TMP/co $R/build/valu_issue >/dev/null 2>&1 || true
fi
CO=$(ls $TMP/*gfx950* 2>/dev/null | head -1)
if [ -z "$CO" ]; then
  objcopy -O binary --only-section=.hip_fatbin $R/build/valu_issue $TMP/fatbin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/valu_issue_gfx950.co
  CO=$TMP/valu_issue_gfx950.co
fi
$OBJDUMP -d --no-show-raw-insn $CO > $TMP/dis.txt
python3 - "$TMP/dis.txt" <<'PY'
import re, sys, collections
funcs, cur = collections.OrderedDict(), None
for line in open(sys.argv[1]):
    m = re.match(r'^[0-9a-f]+ <(\S+)>:\s*$', line)
    if m:
        cur = m.group(1)
        funcs[cur] = []
    elif cur and line.strip():
        funcs[cur].append(line.split('//')[0].strip())
for name, ins in funcs.items():
    m = re.match(r'_Z\d+(k_\w+)Pyi$', name)
    if not m:
        continue
    st = [i for i, l in enumerate(ins) if l.startswith('s_memtime')]
    if len(st) < 2:
        continue
    loop = [l for l in ins[st[0] + 1:st[1]] if not l.startswith('<')]
    ops = collections.Counter(l.split()[0] for l in loop)
    print("== %s: %d instructions between the stamps" % (m.group(1), len(loop)))
    print("   histogram:", dict(ops))
    print("   loop head:", " | ".join(loop[1:4]))
PY
