#!/bin/bash
# Round-3 extra counters asked for by VERDICT r2 (items 3 and 6): one gpurun call, outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/kt.sh tools/brick_bench.py --steps 3 > gpurun_out/brick_bench_kt.txt 2>&1; echo "brick kt $?"
bash tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" brick_accumulate tools/brick_bench.py --steps 3 > gpurun_out/brick_sq_1.txt 2>&1; echo "brick sq1 $?"
bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" brick_accumulate tools/brick_bench.py --steps 3 > gpurun_out/brick_sq_2.txt 2>&1; echo "brick sq2 $?"
bash tools/pmc.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" shade_ tools/wgrad_ab.py --variants 67 --rounds 2 > gpurun_out/shade_mfma_busy.txt 2>&1; echo "shade mfma $?"
