#!/bin/bash
# usage (GPU box): scripts/size_window_matrix.sh > gpurun_out/matrix.jsonl -- bench.py over image sizes x window half widths
# (one JSON line each; the kernel instantiation is in roofline.kernel); SIZES="64 90 ..." restricts the sizes (a call
# of the GPU runner is limited to 20 minutes: three calls of nine sizes), WINDOWS="5 10" the half widths
SIZES=${SIZES:-"64 90 96 100 112 120 128 144 150 160 176 180 192 200 208 224 240 250 256 288 300 320 360 384 400 448 512"}
for n in $SIZES; do
for d in ${WINDOWS:-5 10 13 15 20 30 40}; do
  [ $((2*d+2)) -ge $n ] && continue
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --pixels $n --max-displacement $d --orientations 144 2>/dev/null | tail -1
done; done
