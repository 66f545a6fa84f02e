for cfg in "200 20" "200 30" "200 40" "120 20" "120 30" "120 40" "90 30" "90 40" "100 30" "150 20" "150 30" "150 40" "180 20" "180 25" "180 30" "180 15" "150 15" "60 20" "84 30" "280 30"; do set -- $cfg
python bench.py --pixels $1 --max-displacement $2 --orientations 144 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 +-$2', round(d['value']/1e6,2), d['roofline']['kernel'])"
done
