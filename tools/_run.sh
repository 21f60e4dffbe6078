bash tools/gpu_r3.sh r3g pt2pt
for t in "wave_waves=4"; do
  echo "== $t"; MP2P_HIP_TUNE=$t python tools/wave_probe.py a 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['pose'], d['ms_nn'], d['span_us'], d['dur_us'], 'mean',d['mean_dur']); print('  all  ',d['phase_us_mean']); print('  light',d['light_waves'],d['phase_us_mean_light_waves']); print('  slow', d['slowest(wave, us, nu, rounds, passes, flags)'][:6]); print('  by nu', d['by(count, mean_us, total_ms)']['nu']); print('  last', d['last_to_finish(wave, start_us, dur_us)'][:5])
"
done
bash tools/gpu_r3.sh r3g bencha
