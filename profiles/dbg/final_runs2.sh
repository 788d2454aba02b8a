cd $GRAFT_REPO_ROOT
RG_DIAG=1 timeout 300 python profiles/dbg/seq2_stamps.py 64 0 > gpurun_out/r05B_seq2_stamps.txt 2>&1
head -6 gpurun_out/r05B_seq2_stamps.txt | cut -c1-330
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05B_bench_default.json 2> gpurun_out/r05B_bench_default.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r05B_bench_default.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"))
rf=r["roofline"]; print({k:v for k,v in rf.items() if k not in ("kernel","note")})
for k,v in r["also"].items(): print(k, v.get("ms_per_step"), v.get("value"), v["verified"]["verified"])
PY
bash profiles/dbg/rocprof_bench.sh r05B 2>&1 | tail -32
