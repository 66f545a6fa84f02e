export BIOEM_HIP_LIBRARY=abl/cv4.so
python scripts/parity_shape.py --max-displacement 10
python scripts/dev_parity_check.py g10_n64 g7_n224 2>&1 | grep algo
for w in "--particles 20 --orientations 2304" "--particles 10 --orientations 4608" "--particles 100 --orientations 2304" "--orientations 576"; do for mode in new noxcd; do
  if [ $mode = noxcd ]; then export BIOEM_NO_GROUP_XCD=1; else unset BIOEM_NO_GROUP_XCD; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$w] $mode %.2f M/s  %.3f ms/pass  kernel %.3f ms x %d  %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['kernel']))"
done; done
