#!/bin/bash
# usage (GPU box): bash tools/r03_profiles.sh  -- round 3's bench lines, rocprofv3 kernel statistics and PMC passes kept under profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/final_profiles.sh r03 > gpurun_out/final_r03.log 2>&1
out=gpurun_out/final
bash tools/pmc_dq2.sh a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" > $out/r03_c2_dq_pmc.txt 2>&1
bash tools/pmc_dq2.sh b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" >> $out/r03_c2_dq_pmc.txt 2>&1
bash tools/pmc_traffic.sh r03 c2 > $out/r03_c2_traffic.txt 2>&1
cp gpurun_out/r03_pmc_fetch_size.csv $out/r03_c2_dq_pmc_fetch_size.csv 2>/dev/null
cp gpurun_out/r03_pmc_write_size.csv $out/r03_c2_dq_pmc_write_size.csv 2>/dev/null
( cd /tmp && rm -rf /tmp/prof_c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$out/r03_c3_bench_profiled.json 2>/dev/null )
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $out/r03_c3_kernel_stats.csv
bash tools/timeline_fit.sh c2 > $out/r03_c2_step_gaps.txt 2>&1
timeout -k 10 200 python3 tools/shape_time.py 200000 20000 0.03 200 10 4 3 > $out/r03_c2_matrix_k200.txt 2>/dev/null
ls -la $out
