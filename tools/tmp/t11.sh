mkdir -p gpurun_out/s2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench_steps20.json 2> gpurun_out/s2/bench_steps20.err
python -c "
import json; d=json.loads(open('gpurun_out/s2/bench_steps20.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['utd_matched_us_per_update'], d['cart_sac_env_steps_per_s'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])"
