python -m pytest tests -m gpu -q -k "grad or adam or backward or lbfgs or vfe or composite or kernel_matrix or dimensions" 2>&1 | grep -E "passed|failed|^E  " | tail -5
python bench.py --workload c3 --steps 3 --no-cpu-baseline > gpurun_out/sweep_c3.json 2>/dev/null
python bench.py --workload c2 --steps 20 --no-cpu-baseline > gpurun_out/sweep_c2.json 2>/dev/null
python - <<'PY'
import json
for wl in ("c3", "c2"):
    l = json.loads(open("gpurun_out/sweep_%s.json" % wl).read().strip().splitlines()[-1])
    b = l["loss_backward"]
    print(wl, "bwd ms", round(b["ms_per_step"], 2), {k: round(v, 3) if isinstance(v, float) else v for k, v in b["roofline_grad_sweep"].items() if k in ("achieved", "frac", "avg_launch_us")})
PY
