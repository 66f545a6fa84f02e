cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
( for shp in "--pixels 224 --max-displacement 20" "--pixels 224 --max-displacement 16" "--pixels 224 --max-displacement 23" "--pixels 128 --max-displacement 20" "--pixels 256 --max-displacement 21" "--pixels 208 --max-displacement 18" "--pixels 64 --max-displacement 16" "--pixels 96 --max-displacement 22" "--pixels 160 --max-displacement 19" "--pixels 336 --max-displacement 17"; do
  timeout -k 10 120 python scripts/parity_shape.py $shp 2>&1 | tail -2 | sed "s/^/[$shp] /"
done ) > $O/parity.txt 2>&1
cat $O/parity.txt
( for shp in "--pixels 224 --max-displacement 20" "--pixels 224 --max-displacement 16" "--pixels 128 --max-displacement 20" "--pixels 256 --max-displacement 20" "--pixels 160 --max-displacement 20"; do
  for v in 1 0; do
    for rep in 1 2; do
    if [ $v = 1 ]; then export BIOEM_NO_FASTM2=1; else unset BIOEM_NO_FASTM2; fi
    timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations 288 $shp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$shp] nofastm2=$v %.2f M/s  kernel %.3f ms  %s  check %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['result_check']['ok']))"
    done
  done
done ) > $O/ab.txt 2>&1
cat $O/ab.txt
