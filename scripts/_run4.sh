cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
VARIANTS="fm2_c32 fm2_btab" SHAPES="--max-displacement 20" bash scripts/ab_slim.sh > $O/ab224.txt 2>&1
cat $O/ab224.txt
VARIANTS="fm2_c32_nyq fm2_btab_nyq" SHAPES="--pixels 128 --max-displacement 20" bash scripts/ab_slim.sh > $O/ab128.txt 2>&1
cat $O/ab128.txt
BIOEM_HIP_LIBRARY=abl/fm2_btab.so timeout -k 10 300 bash scripts/pmc_quick.sh fm2btab --max-displacement 20 --orientations 144 > $O/pmcq_btab.txt 2>&1
cat $O/pmcq_btab.txt
