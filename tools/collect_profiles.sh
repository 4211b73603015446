#!/bin/bash
# gpurun_out/ (scratch) -> profiles/<tag>/ (tracked): the summaries of tools/profile_round.sh <tag>.
set -u
TAG=${1:-r06}; P=gpurun_out/prof_$TAG; D=profiles/$TAG
mkdir -p $D
cp $P/kernel_stats.csv $D/final_kernel_stats.csv
cp $P/timed_region_kernel_stats.csv $D/timed_region_kernel_stats.csv
cp $P/pmc_summary.txt $D/final_pmc_summary.txt
for w in struct noise c3x4; do
  f=gpurun_out/pmc_${TAG}_$w/summary.txt
  [ -f $f ] && cp $f $D/$([ $w = struct ] && echo final || echo c2noise | sed "s/c2noise/$([ $w = noise ] && echo c2noise || echo c3x4)/")_pmc_phases.txt
done
for c in c2noise c3x4 c3x1 c4 one4k m4 c5m0 c5m4; do
  [ -f $P/${c}_kernel_stats.csv ] && cp $P/${c}_kernel_stats.csv $D/
  [ -f $P/$c.log ] && grep "ms/step" $P/$c.log > $D/${c}_line.txt
  [ -f $P/${c}_pmc_summary.txt ] && cp $P/${c}_pmc_summary.txt $D/
done
for c in trellis1080 trellis4k sharp1080 sharp4k; do
  [ -f $P/${c}_kernel_stats.csv ] && cut -c1-200 $P/${c}_kernel_stats.csv | head -12 > $D/${c}_kernel_stats.csv
  [ -f $P/${c}_line.txt ] && cp $P/${c}_line.txt $D/
done
[ -f $P/sharp_batches.txt ] && cp $P/sharp_batches.txt $D/
cp gpurun_out/$TAG/final_bench.json $D/final_bench.json
cp gpurun_out/$TAG/exchange_n1.json $D/exchange_n1.json
ls $D
