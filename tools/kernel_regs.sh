#!/bin/bash
# VGPRs / SGPRs / LDS / scratch of every kernel of snf_lib.hip as hipcc compiles it for gfx950 (no GPU needed)
# usage: bash tools/kernel_regs.sh [pattern] [extra hipcc flags]
P=${1:-.}; shift
S=$(mktemp /tmp/snf_lib.XXXXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Wno-unused-value -Wno-align-mismatch -w "$@" \
  --cuda-device-only -S sniffles_amd/csrc/snf_lib.hip -o $S || exit 1
python3 - "$S" "$P" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read(); pat = re.compile(sys.argv[2])
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not pat.search(dn): continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    print(f"{dn[:90]:90s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s} accum_off {g('accum_offset')}")
PY
rm -f $S
