# rocprof evidence for the other BASELINE configurations:
#   bash tools/prof_configs.sh r5 cfg3 cfg4 cfg5 cfg4t   -> gpurun_out/<R>_kernel_stats_<cfg>.md, <R>_pmc_traffic_<cfg>.json, <R>_pmc_sq_<cfg>.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export QT_CFG_ROOFLINE=0
R=$1; shift
for c in "$@"; do
  K=k_spmm; case $c in *t|*tp) K=k_attn;; esac
  rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof_$c -o r --output-format csv -- python3 tools/bench_configs.py $c 10 > gpurun_out/${R}_$c.json 2> gpurun_out/${R}_prof_$c.log
  python profiles/summarize.py gpurun_out/${R}_prof_$c "rocprofv3 --kernel-trace --stats -- python3 tools/bench_configs.py $c 10" > gpurun_out/${R}_kernel_stats_$c.md
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pf_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/${R}_pf_$c.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pw_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/${R}_pw_$c.log
  python profiles/pmc_traffic.py gpurun_out/${R}_pf_$c gpurun_out/${R}_pw_$c "python3 tools/bench_configs.py $c 2" > gpurun_out/${R}_pmc_traffic_$c.json
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/${R}_pg_$c -o r -- python3 tools/bench_configs.py $c 2 > /dev/null 2> gpurun_out/${R}_pg_$c.log
  python profiles/pmc_gemm.py gpurun_out/${R}_pg_$c $K > gpurun_out/${R}_pmc_sq_$c.json
  rm -rf gpurun_out/${R}_prof_$c gpurun_out/${R}_pf_$c gpurun_out/${R}_pw_$c gpurun_out/${R}_pg_$c
  echo "$c done"; head -14 gpurun_out/${R}_kernel_stats_$c.md | tail -8
done
