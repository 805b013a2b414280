#!/bin/bash
# usage: tools/placement_counters.sh OUTDIR   (on the GPU box; cd /tmp && export TMPDIR=/tmp first)
out=$1; mkdir -p $out
python3 tools/placement_counters.py 2>/dev/null | grep "^handle" > $out/event_times.txt
i=0
for group in "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum"; do
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $out/p$i -o pmc -- python3 tools/placement_counters.py > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + '/p[0-9]')):
    rows = []
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows += [r for r in csv.DictReader(open(fn)) if 'k_step2' in r['Kernel_Name']]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(by)
    # 2 rounds x 8 handles x 10 launches, in order
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, i in enumerate(ids):
        h = (k // 10) % 8
        for c, v in by[i].items():
            per[c][h].append(v)
    for c in sorted(per):
        print(c, ' '.join(f'{sum(per[c][h]) / len(per[c][h]):.4g}' for h in range(8)))
PY
