#!/bin/bash
# MFMA utilisation of the pair scorer (north_star: "rocprof must show ... MFMA utilisation for the pairwise scorer").
# One rocprofv3 --pmc pass (SQ counters only) + one kernel-trace pass over the fused edge construction at n = 12 000.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sc_pmc -o p -- python3 tools/debug/exp10.py 12000 1 > gpurun_out/sc_pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/sc_pmc --json gpurun_out/scorer_mfma_pmc.json | grep -i "scorer" | cut -c1-400
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sc_tr -o t -- python3 tools/debug/exp10.py 12000 1 > /dev/null 2>&1 || exit 2
find gpurun_out/sc_tr -name "*kernel_stats.csv" -exec cp {} gpurun_out/scorer_fused_n12000_kernel_stats.csv \;
grep -h "scorer" gpurun_out/scorer_fused_n12000_kernel_stats.csv | sed 's/(.*)"//' | cut -c1-160
rm -rf gpurun_out/sc_pmc gpurun_out/sc_tr
