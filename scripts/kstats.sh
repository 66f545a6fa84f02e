#!/bin/bash
# per-kernel time of one bench workload: scripts/kstats.sh TAG <bench.py arguments>; prints the stats table and leaves it in
# gpurun_out/kstats_TAG/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/kstats_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/bench.log 2>&1
python3 - "$O" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%9.2f ms' % (int(r['TotalDurationNs']) / 1e6),
          '%8.1f us avg' % (float(r['AverageNs']) / 1e3), r['Percentage'].rjust(6))
P
tail -1 $O/bench.log | cut -c1-300
