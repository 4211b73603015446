#!/bin/bash
# Everything a round's profiles/ directory is made of, in one GPU session:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r05'
# -> gpurun_out/prof_<tag>/ (kernel stats, timed-region stats, PMC summary of the bench workload), gpurun_out/pmc_<tag>_*/
# (VALU per wave by ablation level, struct and noise), per-configuration stats, the bench line.
set -u
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
bash tools/profile_gpu.sh $TAG > gpurun_out/$TAG/profile_gpu.log 2>&1
bash tools/pmc_phases.sh ${TAG}_struct "0 1 2 3" > gpurun_out/$TAG/pmc_phases_struct.txt 2>&1
PHASE_CMD="python $PWD/tools/profile_workload.py c2noise 3" bash tools/pmc_phases.sh ${TAG}_noise "0 1 2 3" > gpurun_out/$TAG/pmc_phases_noise.txt 2>&1
PHASE_CMD="python $PWD/tools/profile_workload.py c3x4 3" bash tools/pmc_phases.sh ${TAG}_c3x4 "0 1 2 3" > gpurun_out/$TAG/pmc_phases_c3x4.txt 2>&1
bash tools/profile_configs.sh $TAG "c2noise c3x4 c3x1 c4 one4k m4 c5m0 c5m4" "c3x4 one4k m4" > gpurun_out/$TAG/profile_configs.log 2>&1
python bench.py > gpurun_out/$TAG/final_bench.json 2> gpurun_out/$TAG/final_bench.err
python bench.py --gpus 1 --exchange --no-cpu-baseline --no-other-configs > gpurun_out/$TAG/exchange_n1.json 2> gpurun_out/$TAG/exchange_n1.err
tail -c 600 gpurun_out/$TAG/final_bench.json
