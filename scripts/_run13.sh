VARIANTS="pb2 wf" SHAPES="--max-displacement 40 --envelopes 4 --defocus 8" scripts/ab_slim.sh
VARIANTS="w20_old w20_new" SHAPES="--max-displacement 20" scripts/ab_slim.sh
