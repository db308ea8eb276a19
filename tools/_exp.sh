set -e
mkdir -p gpurun_out/exp2
timeout -k 10 600 python -u -m pytest tests/test_dense_gpu.py tests/test_cmodel_gpu.py tests/test_config_shapes_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/exp2/pytest.log 2>&1 || { tail -30 gpurun_out/exp2/pytest.log; exit 1; }
tail -3 gpurun_out/exp2/pytest.log
for v in "base:" "b8192:--global-batch 8192" "kaggle:--config kaggle"; do
  n=${v%%:*}; f=${v#*:}
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 $f > gpurun_out/exp2/$n.json 2> gpurun_out/exp2/$n.err
  python - <<P
import json
d=json.loads(open("gpurun_out/exp2/$n.json").read().strip().splitlines()[-1])
print("$n", d["value"], d["ms_per_step"])
for k in d["kernels"]: print("   %-34s %.3f ms  %.3f %s frac %.2f"%(k["op"],k["ms_per_step"],k["achieved"],k["unit"],k["frac"]))
P
done
