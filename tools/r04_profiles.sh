#!/bin/bash
# usage (GPU box): bash tools/r04_profiles.sh  -- round 4's bench lines, rocprofv3 kernel statistics and PMC passes kept under profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/final_profiles.sh r04 > gpurun_out/final_r04.log 2>&1
out=gpurun_out/final
bash tools/pmc_dq2.sh a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" > $out/r04_c2_dq_pmc.txt 2>&1
bash tools/pmc_dq2.sh b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" >> $out/r04_c2_dq_pmc.txt 2>&1
bash tools/pmc_traffic.sh r04 c2 > $out/r04_c2_traffic.txt 2>&1
cp gpurun_out/r04_pmc_fetch_size.csv $out/r04_c2_dq_pmc_fetch_size.csv 2>/dev/null
cp gpurun_out/r04_pmc_write_size.csv $out/r04_c2_dq_pmc_write_size.csv 2>/dev/null
( echo "MFMA counters of the dense (tall-skinny QR / small-SVD) kernels, one bench.py step each, rocprofv3 --pmc (tools/pmc_mfma.sh);"
  echo "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); one v_mfma_f64_16x16x4_f64 = 2048 flop; f64 matrix peak 78.6 TFLOP/s"
  for wl in c2 c4 c5; do echo; echo "== $wl"; bash tools/pmc_mfma.sh r04$wl $wl; done ) > $out/r04_mfma_pmc.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_c3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$out/r04_c3_bench_profiled.json 2>/dev/null )
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $out/r04_c3_kernel_stats.csv
bash tools/timeline_fit.sh c2 > $out/r04_c2_step_gaps.txt 2>&1
bash tools/pmc_fill.sh > $out/r04_c2_fill_pmc.txt 2>&1
ls -la $out
