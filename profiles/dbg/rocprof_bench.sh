# rocprofv3 kernel stats of the default bench workload: bash profiles/dbg/rocprof_bench.sh <out-name> [bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=$1; shift
D=gpurun_out/prof_$N
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D --output-format csv -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-also "$@" > $D/bench.json 2> $D/err.txt
tail -1 $D/bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'steady', d.get('steady_state_ms_per_step'), 'verified', d.get('verified'))"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
out=open("gpurun_out/${N}_kernel_stats.txt","w")
print("total kernel time %.1f ms" % (tot/1e6), file=out)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print("%-100s calls=%6s total_ms=%9.2f avg_us=%9.2f pct=%s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]), file=out)
out.close()
print(open("gpurun_out/${N}_kernel_stats.txt").read()[:3500])
PY
# grid sizes of the heavy kernels (CU time = duration x min(grid, 256) CUs)
python3 - <<PY
import csv,glob,collections
f=glob.glob("$D/**/*kernel_trace.csv", recursive=True)[0]
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"][:60]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    wg=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])//max(1,int(r["Workgroup_Size_X"])*int(r["Workgroup_Size_Y"])*int(r["Workgroup_Size_Z"]))
    a=agg[n]; a[0]+=1; a[1]+=d; a[2]+=d*min(wg,256)
tot=sum(a[2] for a in agg.values())
out=open("gpurun_out/${N}_cu_time.txt","w")
print("CU x time by kernel (duration x min(workgroups, 256)); total %.0f CU ms" % tot, file=out)
for n,a in sorted(agg.items(), key=lambda x:-x[1][2])[:25]:
    print("%-62s calls=%6d total_ms=%9.2f cu_ms=%11.0f share=%.3f" % (n,a[0],a[1],a[2],a[2]/tot), file=out)
out.close()
print(open("gpurun_out/${N}_cu_time.txt").read())
PY
find $D -name "*.csv" -size +2M -delete
