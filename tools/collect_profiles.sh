#!/bin/bash
# One gpurun call behind a profiles/ set:   tools/collect_profiles.sh <tag>   (e.g. r03_v1)
# default bench (C4) + the sigma = 0.5 px second run of SURVEY.md 8(d), rocprofv3 kernel trace,
# separate --pmc FETCH_SIZE / WRITE_SIZE passes, C1-C3 + the two off-path configs, pose-only
# latency, the dense sweep beside the rocSOLVER comparator.  Outputs under gpurun_out/<tag>_*.
T=${1:-r03_v1}
O=gpurun_out
export TMPDIR=/tmp
python bench.py > $O/${T}_c4_bench.json 2> $O/${T}_c4_bench.err && \
python bench.py --sigma 0.5 --no-cpu-baseline > $O/${T}_c4_sigma05_bench.json 2> /dev/null && \
rocprofv3 --kernel-trace --stats -d $O/prof_$T --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_$T.log 2>&1 && \
python tools/timeline.py $O/prof_$T > $O/${T}_c4_timeline.txt 2>&1 && \
cp $O/prof_$T/*/*kernel_stats.csv $O/${T}_c4_kernel_stats.csv && rm -f $O/prof_$T/*/*kernel_trace.csv && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$T --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch_$T.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$T --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write_$T.log 2>&1 && \
python tools/pmc_summary.py $O/pmc_fetch_$T $O/pmc_write_$T > $O/${T}_c4_pmc_fetch_write.txt 2>&1 && \
rm -rf $O/pmc_fetch_$T $O/pmc_write_$T && \
for c in C3 C2 C1 C4R W20 DENSE1K; do python bench.py --config $c > $O/${T}_$(echo $c | tr A-Z a-z)_bench.json 2>/dev/null; done
python bench.py --config C4 --stream 4 --steps 10 --warmup 2 > $O/${T}_c4_stream4_bench.json 2>/dev/null
# N > 1 as the driver starts it (plain python): two ranks REHEARSE on the one card (gloo, host-staged)
BA_BENCH_BACKEND=gloo python bench.py --gpus 2 --config C3 --no-cpu-baseline --no-roofline > $O/${T}_c3_gpus2_rehearsal.json 2>/dev/null
BA_BENCH_BACKEND=gloo python bench.py --gpus 2 --weak --config C3 --no-cpu-baseline --no-roofline > $O/${T}_c3_gpus2_weak_rehearsal.json 2>/dev/null
./tools/tile16_bench > $O/${T}_tile16_bench.txt 2>&1
python tools/pose_only_bench.py > $O/${T}_c5_pose_only.txt 2>&1
python tools/dense_bench.py --n 5970 > $O/${T}_dense_sweep.txt 2>&1
tools/rocsolver_bench 5970 5 >> $O/${T}_dense_sweep.txt 2>&1
./cpp/build/test_compare > $O/${T}_c5_compare_autodiff_lm.txt 2>&1
bash tools/sq_counters.sh > $O/${T}_c4_sq_counters.txt 2>&1
echo done
