cd $GRAFT_REPO_ROOT
for n in 1 2 3; do
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|^FAILED" | head -5
done
for o in "" "--opt cam_side=0"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras-frames 40 $o > gpurun_out/r04_z9_bench_driver.json 2>gpurun_out/r04_z9_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_z9_bench_driver.json").read().strip().splitlines()[-1])
c=d["value_sharded"].get("config5_world_of_one") or {}
print("driver [$o]", d["value"], d["value_sharded"].get("value"), c.get("in_frame"), c.get("ahead"), d["value_sharded"].get("error"))
PY
done
