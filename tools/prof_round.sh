cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F="--steps 2 --warmup 3 --frozen-steps 0 --repeats 0 --no-split-variant --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d gpurun_out/r4_prof -o r --output-format csv -- python3 bench.py --steps 20 --warmup 5 --repeats 0 --no-split-variant --no-cpu-baseline > gpurun_out/r4_bench_under_rocprof.json 2> gpurun_out/r4_prof.log
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_pmc_f -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/r4_pmc_f.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_pmc_w -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/r4_pmc_w.log
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/r4_pmc_g -o r -- python3 bench.py $F > /dev/null 2> gpurun_out/r4_pmc_g.log
echo "sq done"
python profiles/summarize.py gpurun_out/r4_prof "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --repeats 0 --no-split-variant --no-cpu-baseline" > gpurun_out/r4_kernel_stats.md
python profiles/pmc_traffic.py gpurun_out/r4_pmc_f gpurun_out/r4_pmc_w > gpurun_out/r4_pmc_traffic.json
python profiles/pmc_gemm.py gpurun_out/r4_pmc_g k_gate_cell > gpurun_out/r4_pmc_gemm.json
python profiles/pmc_gemm.py gpurun_out/r4_pmc_g k_dgrad_cell > gpurun_out/r4_pmc_dgrad.json
python profiles/pmc_gemm.py gpurun_out/r4_pmc_g k_cheb_clip > gpurun_out/r4_pmc_clip_sq.json
rm -rf gpurun_out/r4_prof gpurun_out/r4_pmc_f gpurun_out/r4_pmc_w gpurun_out/r4_pmc_g
head -30 gpurun_out/r4_kernel_stats.md
