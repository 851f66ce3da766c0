R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
(python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pw_bench.json 2>/dev/null) &
BP=$!
for i in $(seq 1 24); do
  sleep 1
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' ' | sed 's/GPU\[0\]//g;s/  */ /g' | cut -c1-300
  echo
done > gpurun_out/pw_smi.log 2>&1
wait $BP
python -c "
import json
d=json.loads(open('gpurun_out/pw_bench.json').read().strip().splitlines()[-1]); print(d['value'])"
sed -n 1,24p gpurun_out/pw_smi.log
