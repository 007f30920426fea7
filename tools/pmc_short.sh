#!/bin/bash
# PMC passes over the short-vector shapes (tools/short_shapes.py): bash tools/pmc_short.sh gpurun_out/dir
# (separate --pmc passes with --kernel-trace only - never combined with the sys / hip / hsa trace domains)
set -u
OUT=${1:-gpurun_out/pmc_short}
mkdir -p "$OUT"
export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
C="FETCH_SIZE TCC_HIT_sum"
D="WRITE_SIZE TCC_MISS_sum TCC_EA0_RDREQ_sum"
E="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAVES"
i=0
for P in "$A" "$B" "$C" "$D" "$E"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc_$i" -- python3 tools/short_shapes.py 1000000 3 > "$OUT/pmc_$i.log" 2>&1 || echo "pmc pass $i failed" >> "$OUT/pmc_fail.log"
done
python3 tools/pmc_summary.py --json "$OUT/pmc_summary.json" "$OUT"/pmc_? > "$OUT/pmc_summary.md"
echo done
