#!/bin/bash
# Device-only compile of one csrc/*.hip file and an instruction histogram of the kernels matching a pattern.
# Usage: tools/asm_hist.sh apd_kernels k67_update_strongILi8ELb1 [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
F=$1; PAT=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only "$@" \
  "$ROOT/apd-mvs_amd/csrc/$F.hip" -o /tmp/$F.s 2>&1 | grep -v "warning: argument unused" | head -20
python3 - "$PAT" /tmp/$F.s <<'PY'
import re, sys
from collections import Counter
pat, path = sys.argv[1], sys.argv[2]
s = open(path).read()
for f in re.split(r'\n(?=_ZN3apd\S+:\s)', s):
    name = f.split(':')[0]
    if pat in name:
        lines = f.split('\n')
        ins = [l.split()[0] for l in lines if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = Counter(ins)
        print(name, len(ins), "instructions")
        print(' '.join('%s:%d' % kv for kv in c.most_common(70)))
        for l in lines:
            if re.search(r'; (ScratchSize|Occupancy|NumVgprs|NumAgprs|codeLenInByte)', l):
                print(l.strip())
PY
