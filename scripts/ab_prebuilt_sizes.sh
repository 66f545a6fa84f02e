for w in "" "--pixels 128" "--pixels 256" "--pixels 225" "--pixels 127" "--max-displacement 15" "--max-displacement 12" "--max-displacement 5" "--pixels 200" "--pixels 320" "--write-angles"; do
for v in $VARIANTS; do cp abl/$v.so bioem_amd/lib/libbioem_hip.so
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --orientations 576 $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$w]', round(d['value']/1e6,2), d['roofline']['kernel'])"
done; done
