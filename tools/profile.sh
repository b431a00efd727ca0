#!/bin/bash
# Round profile on a GPU box (run through gpurun from the repo root): bash tools/profile.sh [round-tag]
#  1. python bench.py (default workload, with the CPU baseline and the extras)  -> gpurun_out/prof/bench.json
#  2. rocprofv3 --kernel-trace --stats of the same command (no CPU baseline)    -> gpurun_out/prof/stats/
#  3. HBM bytes of the kernels: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, on the bench's own tile
#     (the default 32768 sites: the number bench.py reports is measured on this launch shape, not scaled)
#     -> gpurun_out/prof/<tag>_traffic.json in the form bench.py reads from profiles/
# Copy what is to be judged from gpurun_out/prof/ into profiles/.
set -e
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --extras 0 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --extras 0 > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
  echo "$c done"
done
python3 - <<PY
import csv,glob,json,collections
res=collections.OrderedDict()
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$OUT/%s/**/*counter_collection.csv"%c, recursive=True)[0]
    first={}
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not k.startswith("bcfgpu::"): continue
        d=r["Dispatch_Id"]
        if first.setdefault(k,d)!=d: continue
        res.setdefault(k,{}).setdefault(c,0.0)
        res[k][c]+=float(r["Counter_Value"])
json.dump(res, open("$OUT/traffic_raw.json","w"), indent=1)
b=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
g=[k for k in res if "glfgen_kernel" in k][0]
f,w=res[g]["FETCH_SIZE"],res[g]["WRITE_SIZE"]
out={"source":"tools/profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), python bench.py --steps 2 --warmup 1 (the default tile), first dispatch of each kernel",
     "kernel":"glfgen_kernel","sites":b["config"]["sites_per_step_per_gpu"],"samples":b["config"]["samples"],"depth":b["config"]["depth"],
     "fetch_size_kb":f,"write_size_kb":w,
     "correction":"gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is (the kernel's stores are coalesced 4 B per lane, an access width the guide calls uncalibrated)",
     "hbm_bytes_per_launch":int((2*f+w)*1024),
     "algorithmic_bytes_per_launch":b["roofline"]["algorithmic_bytes_per_launch"],
     "other_kernels":{k.replace("bcfgpu::",""):{"fetch_size_kb":v.get("FETCH_SIZE"),"write_size_kb":v.get("WRITE_SIZE")} for k,v in res.items() if k!=g}}
json.dump(out, open("$OUT/${TAG}_traffic.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp $OUT/stats/stats_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || true
tail -1 $OUT/bench.json | cut -c1-600
