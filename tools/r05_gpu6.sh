#!/bin/bash
# round 5, sixth GPU session: the profile set of the round on the final kernel sources (tools/profile.sh), SQ counters incl. LDS bank
# conflicts (tools/sq_all.sh), the bench lines kept for judging (tools/final_set.sh), HBM traffic at four genomes per batch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
bash tools/profile.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -45 gpurun_out/profile_r05.log
bash tools/sq_all.sh > /dev/null 2>&1; head -30 gpurun_out/sq_all/summary.txt | cut -c1-210
bash tools/final_set.sh r05 2>&1 | tail -12
bash tools/pmc.sh --genomes 4 > gpurun_out/pmc_g4.log 2>&1; cp gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic_genomes4.json 2>/dev/null; tail -3 gpurun_out/pmc_g4.log
