mkdir -p gpurun_out
echo "== 4-wave fused"; bash tools/power_probe.sh "python tools/ipa_loop.py --seconds 14" 8 2>&1 | grep -E "Power|sclk|ms per" | head -7
echo "== 8-wave two-launch"; export S2S_IPA_WAVES=82; bash tools/power_probe.sh "python tools/ipa_loop.py --seconds 14" 8 2>&1 | grep -E "Power|sclk|ms per" | head -7
