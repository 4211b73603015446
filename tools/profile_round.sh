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
# the two kinds outside the hot path (round 6): trellis (host API, methods 7 / 8) and the sharp conversion, per-kernel
R=$PWD
for job in "trellis1080 trellis_time.py 1920 1080 10" "trellis4k trellis_time.py 3840 2160 10" "sharp1080 sharp_time.py 1920 1080 10" "sharp4k sharp_time.py 3840 2160 10"; do
  set -- $job
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o s -- python $R/tools/$2 $3 $4 $5 > $R/gpurun_out/$TAG/$1.log 2>&1)
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/prof_$TAG/$1_kernel_stats.csv
  grep "ms  " gpurun_out/$TAG/$1.log > gpurun_out/prof_$TAG/$1_line.txt
done
python tools/sharp_batch_time.py 2>/dev/null | grep batch > gpurun_out/prof_$TAG/sharp_batches.txt
python bench.py > gpurun_out/$TAG/final_bench.json 2> gpurun_out/$TAG/final_bench.err
python bench.py --gpus 1 --exchange --no-cpu-baseline --no-other-configs > gpurun_out/$TAG/exchange_n1.json 2> gpurun_out/$TAG/exchange_n1.err
tail -c 600 gpurun_out/$TAG/final_bench.json
