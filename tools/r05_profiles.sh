#!/bin/bash
# usage (GPU box): bash tools/r05_profiles.sh  -- round 5's bench lines, rocprofv3 kernel statistics and PMC passes kept under profiles/
# (bench.py's own line now carries c3 / c4_1gpu / c5_1gpu; the per-workload runs below are the longer ones behind it)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/final_profiles.sh r05 > gpurun_out/final_r05.log 2>&1
out=gpurun_out/final
bash tools/pmc_dq2.sh a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" > $out/r05_c2_dq_pmc.txt 2>&1
bash tools/pmc_dq2.sh b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" >> $out/r05_c2_dq_pmc.txt 2>&1
for wl in c2 c4 c5; do
  bash tools/pmc_traffic.sh r05$wl $wl > $out/r05_${wl}_traffic.txt 2>&1
  cp gpurun_out/r05${wl}_pmc_fetch_size.csv $out/r05_${wl}_dq_pmc_fetch_size.csv 2>/dev/null
  cp gpurun_out/r05${wl}_pmc_write_size.csv $out/r05_${wl}_dq_pmc_write_size.csv 2>/dev/null
done
( cd /tmp && rm -rf /tmp/prof_c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$out/r05_c3_bench_profiled.json 2>/dev/null )
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $out/r05_c3_kernel_stats.csv
bash tools/timeline_fit.sh c2 > $out/r05_c2_step_gaps.txt 2>&1
bash tools/pmc_fill.sh > $out/r05_c2_fill_pmc.txt 2>&1
ls -la $out
