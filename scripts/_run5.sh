cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout -k 10 500 python scripts/fuzz_wide_windows.py 41 120 > $O/fuzz.txt 2>&1
tail -25 $O/fuzz.txt
