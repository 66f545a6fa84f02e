#!/bin/bash
# usage: BENCH="<bench args>" scripts/ab_build_variants.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
for extra in "$@"; do
  rm -f bioem_amd/lib/libbioem_hip.so
  make -s -C bioem_amd/csrc EXTRA="$extra" all >/dev/null 2>&1 || { echo "build failed: $extra"; continue; }
  echo "== EXTRA [$extra]"
  IFS='|' read -ra sets <<< "$BENCH"
  for a in "${sets[@]}"; do bash scripts/bench_variants.sh "$a"; done
done
