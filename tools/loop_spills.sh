#!/bin/bash
# loop_spills.sh <file.hip> <mangled regex> [flags] -- scratch loads/stores inside the innermost loops of one kernel
tools/isa_of.sh "$@" > /dev/null
python3 - <<'PY'
import re
L=open('/tmp/isa.s').read().split('\n')
hdr=[i for i,l in enumerate(L) if 'Loop Header' in l]
for h in hdr:
    lab=L[h].split(':')[0]
    end=max(i for i,l in enumerate(L) if re.search(r's_cbranch\w+ '+re.escape(lab)+r'\b',l))
    body=L[h:end]
    st=sum('scratch_store' in l for l in body); ld=sum('scratch_load' in l for l in body)
    print(f"loop {lab}: {end-h} lines, {st} scratch stores, {ld} scratch loads, {sum(('v_' in l and '_f64' in l) for l in body)} f64 ops")
tot_st=sum('scratch_store' in l for l in L); tot_ld=sum('scratch_load' in l for l in L)
print(f"kernel: {len(L)} lines, {tot_st} scratch stores, {tot_ld} scratch loads")
PY
