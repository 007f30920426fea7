#!/bin/bash
# The round's profile evidence, on the GPU box:  bash tools/profile_round.sh gpurun_out/r4prof
# (rocprofv3 --kernel-trace --stats of the bench command, of config 5 and of the rerank alone; separate --pmc passes of
#  tools/pmc_target.py and tools/short_shapes.py - never combined with the sys / hip / hsa trace domains; summaries for profiles/)
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {   # stats <name> <command...>: kernel trace + stats of one command, its stdout / stderr beside them
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/${name}_under_rocprof.json" 2> "$OUT/${name}_under_rocprof.err"
  cp "$(ls "$OUT/$name"/*/*kernel_stats.csv | head -1)" "$OUT/${name}_kernel_stats.csv" 2>/dev/null
}
stats bench python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras
python3 tools/trace_steps.py "$OUT/bench" > "$OUT/bench_kernel_trace_timed_steps.json" 2>> "$OUT/bench_under_rocprof.err"
stats c5 python3 bench.py --only c5 --no-check
stats rerank python3 bench.py --only rerank --no-cpu-baseline
stats query python3 bench.py --only query
stats short python3 tools/short_shapes.py 1000000 10
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
C="FETCH_SIZE TCC_HIT_sum"
D="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum"
i=0
for P in "$A" "$B" "$C" "$D"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc_$i" -- python3 tools/pmc_target.py all > "$OUT/pmc_$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/pmc_fail.log"
done
python3 tools/pmc_summary.py --json "$OUT/pmc_summary.json" "$OUT/pmc_1" "$OUT/pmc_2" "$OUT/pmc_3" "$OUT/pmc_4" > "$OUT/pmc_summary.md"
python3 tools/traffic_from_pmc.py "$OUT/pmc_summary.json" > "$OUT/traffic.json"
bash tools/pmc_short.sh "$OUT/pmc_short" > "$OUT/pmc_short.log" 2>&1
echo done
