#!/bin/bash
# Round-end evidence: timing + kernel trace + the three PMC passes for the headline and the C3 / C5 shapes, folded
# into gpurun_out/pmc_traffic.json (copy to profiles/r2/).  Usage: bash tools/prof_all.sh
set -e
rm -f gpurun_out/pmc_traffic.json
bash tools/prof_workload.sh r2_reddit --workload reddit --extras none > gpurun_out/prof_all.log 2>&1
python3 tools/pmc_traffic.py r2_reddit reddit config 6 gpurun_out/pmc_traffic.json
bash tools/prof_workload.sh r2_products_config --workload products --extras none >> gpurun_out/prof_all.log 2>&1
python3 tools/pmc_traffic.py r2_products_config products config 6 gpurun_out/pmc_traffic.json
bash tools/prof_workload.sh r2_products_community --workload products --graph community --extras none >> gpurun_out/prof_all.log 2>&1
python3 tools/pmc_traffic.py r2_products_community products community 6 gpurun_out/pmc_traffic.json
bash tools/prof_workload.sh r2_papers_shard_config --workload papers100M-shard --extras none >> gpurun_out/prof_all.log 2>&1
python3 tools/pmc_traffic.py r2_papers_shard_config papers100M-shard config 6 gpurun_out/pmc_traffic.json
bash tools/prof_workload.sh r2_papers_shard_community --workload papers100M-shard --graph community --extras none >> gpurun_out/prof_all.log 2>&1
python3 tools/pmc_traffic.py r2_papers_shard_community papers100M-shard community 6 gpurun_out/pmc_traffic.json
