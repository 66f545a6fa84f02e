set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/gputest.log
tail -3 gpurun_out/r04a/gputest.log
timeout -k 10 300 python bench.py > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err; tail -c 1500 gpurun_out/r04a/bench_default.json
BIOEM_HIP_LIBRARY=abl/stamps40.so timeout -k 10 200 python scripts/w2_phase_stamps.py 40 > gpurun_out/r04a/stamps40.txt 2>&1
BIOEM_HIP_LIBRARY=abl/stamps20.so timeout -k 10 200 python scripts/w2_phase_stamps.py 20 > gpurun_out/r04a/stamps20.txt 2>&1
cat gpurun_out/r04a/stamps40.txt gpurun_out/r04a/stamps20.txt
timeout -k 10 300 bash scripts/pmc_quick.sh w20 --max-displacement 20 --orientations 144 > gpurun_out/r04a/pmcq_w20.txt 2>&1
cat gpurun_out/r04a/pmcq_w20.txt
