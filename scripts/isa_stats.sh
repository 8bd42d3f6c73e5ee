#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# usage: scripts/isa_stats.sh <file.hip> <kernel-name-regex>   -> registers / LDS / spills and instruction counts of matching kernels (no GPU needed)
set -e
cd "$(dirname "$0")/../asr_amd/csrc"
f=$1; pat=$2
mkdir -p /tmp/isa
extra=""; [ "$f" = rnn.hip ] && extra="-mllvm -amdgpu-kernarg-preload-count=9"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include $extra -x hip --cuda-device-only -S $f -o /tmp/isa/$f.s 2>/dev/null
python3 - "$f" "$pat" <<'PY'
import re, sys
s = open(f"/tmp/isa/{sys.argv[1]}.s").read()
pat = sys.argv[2]
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S):
    a, lds, name, sg, vg, sp = m.groups()
    if re.search(pat, name):
        b = re.search(r"^" + re.escape(name) + r":.*?s_endpgm", s, re.S | re.M).group(0)
        print(name[:100], "| vgpr", vg, "agpr", a, "sgpr", sg, "lds", lds, "spill", sp, "| lines", len(b.splitlines()), "mfma", b.count("v_mfma"),
              "scratch", b.count("scratch_"), "glds", b.count("global_load_lds"), "branches", b.count("s_cbranch"))
PY
