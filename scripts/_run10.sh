for w in "--pixels 512 --max-displacement 40" "--pixels 512 --max-displacement 20" "--pixels 256 --max-displacement 45" "--pixels 448 --max-displacement 20"; do for r in "" 32; do
  if [ -n "$r" ]; then export BIOEM_W2_R=$r; else unset BIOEM_W2_R; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --orientations 144 $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$w] R=$r %.2f M/s  %s' % (d['value']/1e6, d['roofline']['kernel']))"
done; done
