#!/bin/bash
# runs the micro-benchmark block by block (all blocks in one launch fault: a register the blocks share -- not chased)
for m in 1 2 4 8 16 32 64 128 256 512 1024; do
  timeout -k 5 60 tools/ubench/lds_rate $m > /tmp/lr.log 2>&1; rc=$?
  if [ $rc -ne 0 ] || grep -q "fault" /tmp/lr.log; then echo "FAULT at mask $m"; cat /tmp/lr.log; exit 1; fi
  python3 - $m <<'PY'
import sys, re
m = int(sys.argv[1]); k = m.bit_length() - 1
txt = open('/tmp/lr.log').read().split('waves in the workgroup:')[1:]
vals = []
name = None
for blk in txt:
    lines = [l for l in blk.splitlines()[1:] if l.startswith('  ')]
    l = lines[k]
    name = l[2:54].strip(); vals.append(float(l[54:]))
print("%-54s 1 wave %7.2f   2 waves %7.2f   4 waves %7.2f" % (name, vals[0], vals[1], vals[2]))
PY
done
