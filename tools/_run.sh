export MP2P_HIP_TUNE=wave_kernel=1
for n in 16384 131072; do
PROBE_NL=$n python tools/wave_probe.py a chain 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('n=$n', d['pose'], d['ms_nn'], d['span_us'], d['dur_us'], 'mean',d['mean_dur']); print('  all  ',d['phase_us_mean']); print('  light',d['light_waves'],d['phase_us_mean_light_waves'])
"
done
