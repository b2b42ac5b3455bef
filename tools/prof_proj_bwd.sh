cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export QT_CFG_ROOFLINE=0
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/r5c_pg_pb -o r -- python3 tools/bench_configs.py cfg4t 2 > /dev/null 2> gpurun_out/r5c_pg_pb.log
python profiles/pmc_gemm.py gpurun_out/r5c_pg_pb k_proj_bwd > gpurun_out/r5c_pmc_proj_bwd.json
python profiles/pmc_gemm.py gpurun_out/r5c_pg_pb k_gemm_fwd > gpurun_out/r5c_pmc_gemm_fwd_cfg4t.json
python profiles/pmc_gemm.py gpurun_out/r5c_pg_pb k_gemm_wgrad_group > gpurun_out/r5c_pmc_wgrad_cfg4t.json
rm -rf gpurun_out/r5c_pg_pb
cat gpurun_out/r5c_pmc_proj_bwd.json
