export TMPDIR=/tmp
python bench.py > gpurun_out/bench_c4_v17.json 2> gpurun_out/bench_c4_v17.err && \
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v17 --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_v17.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_v17 --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch_v17.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_v17 --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write_v17.log 2>&1 && \
python bench.py --config C3 --no-cpu-baseline > gpurun_out/bench_c3_v17.json 2>/dev/null && python bench.py --config C2 --no-cpu-baseline > gpurun_out/bench_c2_v17.json 2>/dev/null && python bench.py --config C1 --no-cpu-baseline > gpurun_out/bench_c1_v17.json 2>/dev/null; python tools/pose_only_bench.py > gpurun_out/pose_only_v17.log 2>&1; echo done
