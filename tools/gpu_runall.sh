python tools/gpu_debug.py 2 2>&1 | grep -v amdgpu.ids | tail -1
python tools/gpu_debug.py 3 2>&1 | grep -v amdgpu.ids | tail -1
python tools/gpu_debug.py 4 2>&1 | grep -v amdgpu.ids | tail -1
python tools/gpu_matchcheck.py 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py --no-cpu --steps 5 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/b.json
python - <<'PY'
import json
r=json.load(open("gpurun_out/b.json"))
print("lznt1", r["value"], r["ms_per_step"], r["roofline"]["kernels_ms_per_step"])
for k,v in r["extra"].items(): print(k, v["MB_per_s"], v["ms_per_step"], v["roofline"]["kernels_ms_per_step"])
PY
