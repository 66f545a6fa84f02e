#!/bin/bash
# usage (GPU box): VARIANTS="base new" scripts/ab_prebuilt_windows.sh  -- A/B of prebuilt libraries abl/<variant>.so over the
# wide-window shapes (same device for every arm); leaves the LAST variant installed
for w in "--max-displacement 40 --envelopes 4 --defocus 8" "--max-displacement 16" "--max-displacement 20" "--max-displacement 24" "--max-displacement 30" "--max-displacement 40 --pixels 128" "--max-displacement 15 --pixels 128" "--max-displacement 30 --pixels 128" "--max-displacement 15 --pixels 256" "--max-displacement 24 --pixels 256" "--max-displacement 15 --pixels 160" "--max-displacement 25 --pixels 120" "--max-displacement 20 --pixels 96"; do
for v in $VARIANTS; do cp abl/$v.so bioem_amd/lib/libbioem_hip.so
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations 288 $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$w]', round(d['value']/1e6,2), d['roofline']['kernel'])"
done; done
