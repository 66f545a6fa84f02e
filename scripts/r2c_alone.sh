#!/bin/bash
# kernel times of the r2c alone, fast transform against exact DFT: scripts/r2c_alone.sh "224 384" "128 768" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do for mode in fft dft; do
  O=$R/gpurun_out/r2c_alone_${mode}_${w// /_}
  mkdir -p $O
  BIOEM_R2C=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/scripts/r2c_alone.py $w > $O/log 2>&1
  python3 - "$O" "$w" "$mode" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'r2c' in r['Name'] or 'dft' in r['Name']:
        print(sys.argv[2], sys.argv[3], r['Name'][:60].ljust(60), r['Calls'].rjust(4), '%8.1f us avg' % (float(r['AverageNs']) / 1e3), '%8.1f us min' % (float(r['MinNs']) / 1e3))
P
done; done
