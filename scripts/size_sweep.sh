#!/bin/bash
# usage (GPU box): scripts/size_sweep.sh > gpurun_out/size_sweep.jsonl  -- one bench.py JSON line per configuration
for n in 64 96 99 100 120 127 128 135 150 160 180 192 200 224 225 240 250 256 300 320 384 512; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --pixels $n --orientations 1152 2>/dev/null | tail -1
done
for w in "--max-displacement 5" "--max-displacement 10 --grid 2" "--max-displacement 12" "--max-displacement 15" \
         "--max-displacement 20 --grid 2" "--max-displacement 16" "--max-displacement 20" "--max-displacement 24" \
         "--max-displacement 30" "--max-displacement 40" \
         "--max-displacement 40 --envelopes 4 --defocus 8" "--max-displacement 40 --pixels 128" \
         "--max-displacement 40 --pixels 256" "--max-displacement 15 --pixels 128" "--max-displacement 20 --pixels 128" \
         "--max-displacement 30 --pixels 128" "--max-displacement 15 --pixels 256" "--max-displacement 15 --pixels 160" \
         "--write-angles" "--pixels 225"; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --orientations 576 $w 2>/dev/null | tail -1
done
