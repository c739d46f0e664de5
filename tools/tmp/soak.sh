mkdir -p gpurun_out/s2
for W in cart_ddpg cart_sac pen_ddpg pen_sac; do timeout 600 python tools/soak.py $W 200000 2>&1 | tail -2; done | tee gpurun_out/s2/soak.log
