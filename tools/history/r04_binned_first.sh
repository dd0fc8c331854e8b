# first numbers of the binned K1 form at configs[3]
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04_binned; mkdir -p $out; cd $root
python tools/ab_tune.py --rounds 2 --out $out/k1only.txt -v "stream:" -v "binned:--tune k1_form=binned" -- --config c4 --k1-only --steps 12 --warmup 3 > $out/k1only.log 2>&1; tail -3 $out/k1only.log
python tools/ab_tune.py --rounds 2 --out $out/step.txt -v "stream:" -v "binned:--tune k1_form=binned" -v "binned_nolds:--tune k1_form=binned --tune k1_share_lds_kb=-1" -v "binned_stiles:--tune k1_form=binned --tune k1_share_lds_kb=-1 --tune proj_tile=22 --tune proj_bk=16 --tune tn_tile=22 --tune xcd_order=1" -- --config c4 --steps 20 --warmup 4 > $out/step.log 2>&1; tail -5 $out/step.log
