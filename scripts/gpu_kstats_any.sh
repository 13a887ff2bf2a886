#!/bin/bash
# rocprofv3 kernel-trace stats of any probe script (usage: gpu_kstats_any.sh tag script.py args...); top kernels by total time
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
tag=$1; shift; script=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ka_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka_$tag -o p -- python3 $R/$script "$@" > $O/kany_$tag.log 2>&1
echo "kstats $tag rc=$?"
python3 - "$tag" <<'PY'
import csv,sys,glob
tag=sys.argv[1]
f=glob.glob(f'/tmp/ka_{tag}/**/p_kernel_stats.csv',recursive=True)[0]
rows=[(r["Name"][:90], int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f))]
rows.sort(key=lambda x:-x[2])
out=open(f'/root/repo/gpurun_out/kany_{tag}.txt','w')
for n,c,t,a in rows[:40]:
    line=f"{n:90s} calls={c:6d} total_ms={t:10.2f} avg_us={a:10.1f}"
    print(line); out.write(line+"\n")
PY
