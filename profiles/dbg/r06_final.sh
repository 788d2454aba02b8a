# round 6, final artefacts on the last tree: bash profiles/dbg/r06_final.sh [part ...]   (parts: stamps pmc bench prof stress)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
PARTS=${@:-"stamps pmc bench prof stress"}
for part in $PARTS; do
case $part in
stamps)
  RG_DIAG=1 STAMPS_JSON=gpurun_out/r06_seq2_stamps.json timeout 300 python profiles/dbg/seq2_stamps.py 64 0 > gpurun_out/r06_seq2_stamps.txt 2>&1
  head -3 gpurun_out/r06_seq2_stamps.txt | cut -c1-300 ;;
pmc)
  for form in wide pairs; do
    if [ $form = wide ]; then export SEQ_PAIRS=0 SEQ_DUO=1; else export SEQ_PAIRS=1 SEQ_DUO=1; fi
    D=gpurun_out/pmc_r06_$form; rm -rf $D; mkdir -p $D
    rocprofv3 --kernel-trace --stats -d $D/seq_stats --output-format csv -- python3 profiles/pmc_seq.py > $D/info.txt 2> $D/e1.txt
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/seq_fetch --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e2.txt
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $D/seq_write --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e3.txt
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $D/seq_mfma --output-format csv -- python3 profiles/pmc_seq.py > /dev/null 2> $D/e4.txt
    python3 profiles/pmc_seq_summarize.py $D "$(grep '^{' $D/info.txt | tail -1)" gpurun_out/r06_pmc_seq2_$form.json > gpurun_out/r06_pmc_seq2_$form.txt 2>&1
    tail -3 gpurun_out/r06_pmc_seq2_$form.txt
    find $D -name "*.csv" -size +2M -delete
  done ;;
bench)
  timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/r06_bench_default.json").read().strip().splitlines()[-1])
print({k:r[k] for k in ("value","ms_per_step","steady_state_ms_per_step","verified") if k in r}, (r.get("batch_latency_ms") or {}).get("median"))
rf=r["roofline"]; print({k:v for k,v in rf.items() if k not in ("kernel","note")})
for k,v in r["also"].items(): print(k, v.get("ms_per_step"), v.get("value"), v["verified"]["verified"])
print(r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
PY
  ;;
prof)
  bash profiles/dbg/rocprof_bench.sh r06 2>&1 | tail -30 ;;
stress)
  timeout 900 python profiles/race_stress.py --reps 10 --batches 12 --B 16 --layers 8 --db 4096 --tag r06_full_depth 2>&1 | python profiles/dbg/stress_fmt.py | tail -2
  cp gpurun_out/race_stress_r06_full_depth.json gpurun_out/r06_race_stress_full_depth.json ;;
esac
done
