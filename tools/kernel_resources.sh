#!/bin/bash
# VGPRs / LDS / scratch of the kernels whose (mangled) names match $1 in the built library (no GPU needed).
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$(dirname "$0")/../remixt_amd/libremixt_hip.so"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - "$T/notes.txt" "${1:-.}" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for b in t.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b)
    if not name or not pat.search(name.group(1)):
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    print('%-90s vgpr %4s  agpr %4s  sgpr %4s  lds %6s  scratch %5s' % (name.group(1)[:90], g('vgpr_count'), re.match(r':\s+(\d+)', b).group(1), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
PY
rm -rf $T
