python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python - <<'PY'
import json
j=json.loads([x for x in open("gpurun_out/r06_bench_default.json") if x.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["train2d_f32_mfma"]["value"], j["train3d"]["value"], j["infer"]["value"], j["infer"]["e2e"]["mpixels_s"], j["infer_f32_mfma"]["value"])
e=j["train_e2e"]; print(e["value"], e["iteration_ms_p95"], e["iteration_ms_max"], e["loader_wait_ms"])
r=j["roofline"]; print(r["achieved"], r["frac"], r["step_mfma_frac"], r["avg_launch_ms"])
PY
