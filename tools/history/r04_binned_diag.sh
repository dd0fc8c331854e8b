# where the binned form's time goes at configs[3]: per-kernel durations, fabric-side traffic, L2 hit rate
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
A="--config c4 --k1-only --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --tune k1_form=binned"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python3 $root/bench.py $A > /dev/null 2>&1
python3 - $out/ks <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pool' in r['Name']: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
rm -rf $out/ks
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o f -- python3 $root/bench.py $A > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o w -- python3 $root/bench.py $A > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/ph -o h -- python3 $root/bench.py $A > /dev/null 2>&1
python3 - $out <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pf","pw","ph"):
    for f in glob.glob(sys.argv[1]+"/"+d+"/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            n=row["Kernel_Name"]
            if "pool" in n:
                acc[n.split("<")[0].split("::")[-1]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,c in acc.items():
    m={n: sum(v)/len(v) for n,v in c.items()}
    print(k, {n: (round(v*1024*(2 if n=="FETCH_SIZE" else 1)/1e9,3) if "SIZE" in n else round(v/1e6,2)) for n,v in m.items()}, "(SIZE in GB, FETCH x2; TCC in M requests)")
PY
rm -rf $out/pf $out/pw $out/ph
