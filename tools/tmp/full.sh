mkdir -p gpurun_out/s2
python bench.py --no-cpu-baseline --steps 2000 --warmup 200 2>/dev/null > gpurun_out/s2/full.json; python - <<EOF
import json
d=json.loads(open("gpurun_out/s2/full.json").read().strip().splitlines()[-1])
for k,v in d.items():
    if k not in ("roofline","config","cpu_baseline"): print(k, v)
r=d["roofline"]
print({k:v for k,v in r.items() if k!="all_kernels"})
for k,v in r["all_kernels"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
EOF
