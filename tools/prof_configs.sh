# rocprof evidence for the other BASELINE configurations (the message aggregate on frames of several base cells):
#   bash tools/prof_configs.sh cfg3 cfg5      -> gpurun_out/r4_kernel_stats_<cfg>.md, r4_pmc_traffic_<cfg>.json, r4_pmc_spmm_sq_<cfg>.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export QT_CFG_ROOFLINE=0
for c in "$@"; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof_$c -o r --output-format csv -- python3 tools/bench_configs.py $c 10 > gpurun_out/r4_$c.json 2> gpurun_out/r4_prof_$c.log
  python profiles/summarize.py gpurun_out/r4_prof_$c "rocprofv3 --kernel-trace --stats -- python3 tools/bench_configs.py $c 10" > gpurun_out/r4_kernel_stats_$c.md
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_pf_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/r4_pf_$c.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_pw_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/r4_pw_$c.log
  python profiles/pmc_traffic.py gpurun_out/r4_pf_$c gpurun_out/r4_pw_$c "python3 tools/bench_configs.py $c 2" > gpurun_out/r4_pmc_traffic_$c.json
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/r4_pg_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/r4_pg_$c.log
  python profiles/pmc_gemm.py gpurun_out/r4_pg_$c k_spmm > gpurun_out/r4_pmc_spmm_sq_$c.json
  rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r4_pt_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/r4_pt_$c.log
  python tools/pmc_kernels.py gpurun_out/r4_pt_$c k_spmm > gpurun_out/r4_pmc_spmm_ta_$c.txt 2>&1
  rm -rf gpurun_out/r4_prof_$c gpurun_out/r4_pf_$c gpurun_out/r4_pw_$c gpurun_out/r4_pg_$c gpurun_out/r4_pt_$c
  echo "$c done"; head -14 gpurun_out/r4_kernel_stats_$c.md | tail -8
done
