#!/bin/bash
# scripts/isa_stats.sh <kernel-name-regex> [EXTRA_HIPFLAGS]: device ISA of csrc/fes_api.hip (gfx950), resource usage and an
# instruction histogram of every kernel whose mangled name matches.  Runs without a GPU.
cd "$(dirname "$0")/../fusion-sim_amd" || exit 1
OUT=${ISA_OUT:-/tmp/isa}; mkdir -p $OUT
SRC=${ISA_SRC:-csrc/fes_api.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -munsafe-fp-atomics -Wno-unused-function --cuda-device-only -S $SRC -o $OUT/dev.s -Rpass-analysis=kernel-resource-usage $2 2> $OUT/res.txt || { grep -E "error" -A5 $OUT/res.txt | head -40; exit 1; }
python3 - "$1" $OUT <<'PY'
import re, sys, collections
pat, out = sys.argv[1], sys.argv[2]
s = open(out + '/dev.s').read()
res = open(out + '/res.txt').read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\s*\.size\s+\1', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not re.search(pat, name): continue
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.', '//')) and not l.strip().endswith(':')]
    c = collections.Counter(l.split()[0] for l in lines)
    r = re.search(r'Function Name: ' + re.escape(name) + r'.*?LDS Size[^\n]*', res, re.S)
    rs = dict(re.findall(r'remark:\s+([\w \[\]/]+): (\w+)', r.group(0))) if r else {}
    print(name)
    print('   instructions %d  VGPRs %s  SGPRs %s  scratch %s B/lane  occupancy %s' % (len(lines), rs.get('VGPRs'), rs.get('TotalSGPRs'), rs.get('ScratchSize [bytes/lane]'), rs.get('Occupancy [waves/SIMD]')))
    keys = ['scratch_', 's_swappc', 'v_mad_u64_u32', 'v_mul_lo_u32', 'v_mov_b32', 'ds_read', 'ds_add', 'ds_write', 'global_load', 'global_store', 'global_atomic', 'v_pk_', 'v_fma', 'v_cndmask', 's_cbranch']
    print('   ' + '  '.join('%s %d' % (k, sum(v for n, v in c.items() if n.startswith(k))) for k in keys))
PY
