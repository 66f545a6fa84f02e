cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
VARIANTS="fm2_base fm2_wavebar fm2_c32 fm2_c32w" SHAPES="--max-displacement 20" bash scripts/ab_slim.sh > $O/ab.txt 2>&1
cat $O/ab.txt
BIOEM_HIP_LIBRARY=abl/fm2_base.so timeout -k 10 300 bash scripts/pmc_quick.sh fm2base --max-displacement 20 --orientations 144 > $O/pmcq_base.txt 2>&1
cat $O/pmcq_base.txt
BIOEM_HIP_LIBRARY=abl/fm2_c32w.so timeout -k 10 300 bash scripts/pmc_quick.sh fm2c32w --max-displacement 20 --orientations 144 > $O/pmcq_c32w.txt 2>&1
cat $O/pmcq_c32w.txt
