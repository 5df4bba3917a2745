#!/bin/bash
# kernel time vs input data (power limiting?): rocprofv3 kernel stats of tools/probe_power.py per data kind
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kind in randn zeros ones; do
  rm -rf /tmp/pp_$kind
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$kind -- python3 tools/probe_power.py $kind > /dev/null 2>&1 || { echo "failed $kind"; exit 1; }
  echo "== $kind"
  python3 - /tmp/pp_$kind <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(sorted(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'))[-1])):
    if 'attn_fwd_persistent_kernel' in r['Name'] or 'attn_fwd_x3_kernel' in r['Name']:
        print("  %-52s calls %s avg %.1f us" % (r['Name'][:52], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
