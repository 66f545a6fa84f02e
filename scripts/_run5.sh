BIOEM_FASTM_ALL=1 BIOEM_NO_FASTM=1 VARIANTS="fm10_32" SHAPES="--max-displacement 10" scripts/ab_slim.sh
BIOEM_FASTM_ALL=1 BIOEM_FASTM_R32=1 VARIANTS="fm10_32" SHAPES="--max-displacement 10" scripts/ab_slim.sh
BIOEM_FASTM_ALL=1 VARIANTS="fm10_16" SHAPES="--max-displacement 10" scripts/ab_slim.sh
VARIANTS="fm13" SHAPES="--max-displacement 13" scripts/ab_slim.sh
VARIANTS="fm15" SHAPES="--max-displacement 15" scripts/ab_slim.sh
