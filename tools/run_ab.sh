for p in planes f16 planes f16; do
S2S_IPA_PATH=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p', round(d['value'],3), round(d['ms_per_step'],1), round(d['roofline']['mean_launch_ms'],3), round(d['ipa_kernel']['mean_launch_ms'],4), round(d['ipa_kernel']['frac'],4))"
done
