run() { # N d R
env BIOEM_W2_R=$3 python bench.py --pixels $1 --max-displacement $2 --orientations 144 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 +-$2 R=$3', round(d['value']/1e6,2), d['roofline']['kernel'])"
}
for d in 20 30 40; do for r in 8 20 10; do run 200 $d $r; done; done
for d in 20 30 40; do for r in 8 30 20 12 10; do run 120 $d $r; done; done
for d in 30 40; do for r in 30 18 10; do run 90 $d $r; done; done
for d in 30 40; do for r in 20 10; do run 100 $d $r; done; done
for d in 20 30 40; do for r in 30 10; do run 150 $d $r; done; done
for d in 20 30 40; do for r in 30 20 18 12 10; do run 180 $d $r; done; done
for d in 20 30; do for r in 10; do run 250 $d $r; done; done
for d in 20 30 40; do for r in 16 8; do run 240 $d $r; done; done
for d in 20 30 40; do for r in 16 8; do run 208 $d $r; done; done
