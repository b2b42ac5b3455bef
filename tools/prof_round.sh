# rocprof evidence of the round's final build for the headline workload (bench.py, BASELINE configs[1]):
#   bash tools/prof_round.sh [r5]   -> gpurun_out/<R>_kernel_stats.md, _pmc_traffic.json, _step_summary.json, _pmc_{gemm,dgrad,clip_sq,remesh_sq}.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r5}
F="--steps 2 --warmup 3 --frozen-steps 0 --repeats 0 --no-split-variant --no-cpu-baseline --no-roofline"
CMD="python3 bench.py --steps 20 --warmup 5 --repeats 0 --no-split-variant --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_prof -o r --output-format csv -- $CMD > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/${R}_prof.log
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_f -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/${R}_pmc_f.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_w -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/${R}_pmc_w.log
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/${R}_pmc_g -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/${R}_pmc_g.log
echo "sq done"
python profiles/summarize.py gpurun_out/${R}_prof "rocprofv3 --kernel-trace --stats -- $CMD" > gpurun_out/${R}_kernel_stats.md
python profiles/pmc_traffic.py gpurun_out/${R}_pmc_f gpurun_out/${R}_pmc_w > gpurun_out/${R}_pmc_traffic.json
python profiles/step_summary.py gpurun_out/${R}_prof gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_step_summary.json
python profiles/pmc_gemm.py gpurun_out/${R}_pmc_g k_gate_cell > gpurun_out/${R}_pmc_gemm.json
python profiles/pmc_gemm.py gpurun_out/${R}_pmc_g k_dgrad_cell > gpurun_out/${R}_pmc_dgrad.json
python profiles/pmc_gemm.py gpurun_out/${R}_pmc_g k_cheb_clip > gpurun_out/${R}_pmc_clip_sq.json
python profiles/pmc_gemm.py gpurun_out/${R}_pmc_g k_remesh_clip > gpurun_out/${R}_pmc_remesh_sq.json
f=$(find gpurun_out/${R}_prof -name "*kernel_trace.csv" | head -1); python tools/trace_sequence.py $f > gpurun_out/${R}_sequence.txt
rm -rf gpurun_out/${R}_prof gpurun_out/${R}_pmc_f gpurun_out/${R}_pmc_w gpurun_out/${R}_pmc_g
head -30 gpurun_out/${R}_kernel_stats.md; cat gpurun_out/${R}_step_summary.json
