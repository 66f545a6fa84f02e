for v in $VARIANTS; do for i in 1 2; do
BIOEM_HIP_LIBRARY=abl/$v.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations 288 $SHAPE 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v %.2f M/s  kernel %.3f ms  %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
done; done
