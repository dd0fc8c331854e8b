# configs[3]: the projection chain alone (frozen-feature cache inputs) against the chain behind the gather on one stream, kernel by kernel
root=$GRAFT_REPO_ROOT; cd $root
bash tools/kstats.sh c4_cache --no-secondary --config c4 --inputs cache --steps 100 --warmup 10
bash tools/kstats.sh c4_onestream --no-secondary --config c4 --no-overlap --steps 60 --warmup 6
bash tools/kstats.sh c4_onestream_stiles --no-secondary --config c4 --no-overlap --steps 60 --warmup 6 --tune proj_tile=22 --tune proj_bk=16 --tune tn_tile=22 --tune xcd_order=1
for t in c4_cache c4_onestream c4_onestream_stiles; do echo "## $t"; python3 -c "
import json; d=json.load(open('gpurun_out/ks_$t.json')); print('step', round(d['ms_per_step']*1e3,1), 'us')"; cat gpurun_out/ks_$t.txt | cut -c1-60,100-; done
