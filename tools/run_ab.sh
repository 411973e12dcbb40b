export S2S_BENCH_BACKEND=gloo
mkdir -p gpurun_out/ab
for cfg in cfg2 cfg3 cfg5; do
  extra="--replicas 8 --denoise-steps 4"
  [ $cfg = cfg2 ] && extra="--replicas 8 --n-res 64 --denoise-steps 4"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config $cfg --steps 1 --warmup 0 --no-cpu-baseline $extra 2> gpurun_out/ab/dist_$cfg.err | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', l['n_gpus'], round(l['value'],2), l['distributed'])" || tail -5 gpurun_out/ab/dist_$cfg.err
done
