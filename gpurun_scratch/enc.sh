timeout 600 python -m pytest tests -m gpu -x -q -k "encoding or layout or micro or dropin or smoke" 2>&1 | tail -3
for b in 10240 20480 40960; do
OC_ENC_LDS=$b python bench.py --no-cpu-baseline --steps 400 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['encode']; print('$b', 'u8 %.1f us %.2f TB/s | f32 %.1f us %.2f TB/s' % (d['u8']['launch_ms']*1e3, d['u8']['achieved_GBs']/1e3, d['f32']['launch_ms']*1e3, d['f32']['achieved_GBs']/1e3))"
done
