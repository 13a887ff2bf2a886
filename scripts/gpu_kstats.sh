#!/bin/bash
# rocprofv3 kernel-trace stats of the RLGC probe, k_* kernels only (usage: gpu_kstats.sh tag [Z Y X iters]); env knobs pass through
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
tag=${1:-run}; Z=${2:-64}; Y=${3:-2048}; X=${4:-2048}; IT=${5:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o p -- python3 $R/scripts/gpu_probe_rlgc.py $Z $Y $X $IT > $O/kstats_$tag.log 2>&1
echo "kstats $tag rc=$?"
python3 - "$tag" <<'PY'
import csv,sys,re,glob
tag=sys.argv[1]
f=glob.glob(f'/tmp/ks_{tag}/**/p_kernel_stats.csv',recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+(?:<[^>]*>)?)", r["Name"])
    if m: rows.append((m.group(1), int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
rows.sort(key=lambda x:-x[2])
out=open(f'/root/repo/gpurun_out/kstats_{tag}.txt','w')
for n,c,t,a in rows:
    line=f"{n:40s} calls={c:5d} total_ms={t:9.3f} avg_us={a:9.1f}"
    print(line); out.write(line+"\n")
PY
