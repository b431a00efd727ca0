#!/bin/bash
# Round profile on a GPU box (run through gpurun from the repo root): bash tools/profile.sh
#  1. python bench.py (default workload, with the CPU baseline)              -> gpurun_out/prof/bench.json
#  2. rocprofv3 --kernel-trace --stats of the same command (no CPU baseline) -> gpurun_out/prof/stats/
#  3. HBM bytes of the kernels: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, 4096-site tile
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 $R/bench.py --cpu-seconds 0 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --sites 4096 --cpu-seconds 0 > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
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
print(json.dumps(res, indent=1))
PY
cat $OUT/bench.json
