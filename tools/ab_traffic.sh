#!/bin/bash
# A/B of library builds (tools/ab_build.py) on the PMC target: kernel durations and L2 fetch traffic per build.
#   bash tools/ab_traffic.sh gpurun_out/abt ablib/lib_base.so ablib/lib_xnt.so
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  export LSHRS_HIP_LIBRARY="$PWD/$lib"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_stats" -- python3 tools/pmc_target.py all > "$OUT/${name}_stats.log" 2>&1
  cp "$(ls "$OUT/${name}_stats"/*/*kernel_stats.csv | head -1)" "$OUT/${name}_kernel_stats.csv"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$OUT/${name}_pmc" -- python3 tools/pmc_target.py all > "$OUT/${name}_pmc.log" 2>&1
  python3 tools/pmc_summary.py "$OUT/${name}_pmc" > "$OUT/${name}_pmc.md"
  rm -rf "$OUT/${name}_pmc" "$OUT/${name}_stats"
  echo "== $name"; grep -E "sig16_kernel|sig_fix8" "$OUT/${name}_kernel_stats.csv" | cut -c1-200; grep -E "sig16_kernel" "$OUT/${name}_pmc.md"
done
