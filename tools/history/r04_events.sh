#!/bin/bash
# event release scope (trainer fence/ready events + engine fork/join events) and the column-sliced K1 on a table that
# streams from HBM: parity first, then interleaved A/Bs on one box
O=gpurun_out/r04_events; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_model_api.py -x -q -m gpu -k "sliced or golden or ordered or copy_stream or token_store or trainer" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/ab_tune.py --rounds 3 --out $O/c2_events.txt \
  -v "ev0:--event-scope 0" -v "ev1:--event-scope 1 --tune event_scope=1" -v "ev2:--event-scope 2 --tune event_scope=2" \
  -v "ev1_fused:--event-scope 1 --tune event_scope=1 --tune backward=fused" -v "ev1_fork:--event-scope 1 --tune event_scope=1 --tune backward=fork" \
  -- --steps 200 --warmup 20 2>&1 | tail -7
python tools/ab_tune.py --rounds 2 --out $O/c3_events.txt \
  -v "ev0:--event-scope 0" -v "ev1:--event-scope 1 --tune event_scope=1" -v "ev1_fork:--event-scope 1 --tune event_scope=1 --tune backward=fork" \
  -v "ev1_sim60:--event-scope 1 --tune event_scope=1 --simulate-allreduce-us 60" -v "ev0_sim60:--event-scope 0 --simulate-allreduce-us 60" \
  -- --config c3 --steps 200 --warmup 20 2>&1 | tail -7
python tools/ab_tune.py --rounds 2 --out $O/c4_cols.txt \
  -v "stream:" -v "columns:--tune k1_form=columns" -v "cols_t22bk16:--tune k1_form=columns --tune proj_tile=22 --tune proj_bk=16 --tune tn_tile=22 --tune xcd_order=1" \
  -v "cols_nolds:--tune k1_form=columns --tune k1_share_lds_kb=-1 --tune proj_tile=22 --tune proj_bk=16 --tune tn_tile=22 --tune xcd_order=1" \
  -v "stream_ev1:--event-scope 1 --tune event_scope=1" \
  -- --config c4 --steps 30 --warmup 8 2>&1 | tail -7
