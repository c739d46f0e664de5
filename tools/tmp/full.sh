mkdir -p gpurun_out/s2
python bench.py > gpurun_out/s2/bench_default.json 2> gpurun_out/s2/bench_default.err
tail -c 600 gpurun_out/s2/bench_default.json
