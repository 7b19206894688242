for v in base nt unroll; do
for b in 40960 20480; do
OC_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_scratch/encabl_$v.so OC_ENC_LDS=$b python bench.py --no-cpu-baseline --steps 400 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['encode']; print('$v $b', 'u8 %.1f us %.2f TB/s | f32 %.1f us %.2f TB/s' % (d['u8']['launch_ms']*1e3, d['u8']['achieved_GBs']/1e3, d['f32']['launch_ms']*1e3, d['f32']['achieved_GBs']/1e3))"
done; done
