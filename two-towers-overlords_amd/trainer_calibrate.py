"""TripletTrainer.calibrate — timing the step's equivalent schedules on the device at hand ("Engine: slots,
streams, schedules").  A mixin: the method body is TripletTrainer's, kept in a file of its own."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from .trainer_types import Batch, as_batch


class CalibrationMixin:
    INFINITY_CACHE_BYTES = 256 << 20  # MI355X MALL (MI355X_MICROARCH.md): a table beyond it really streams from HBM

    def calibrate(self, batch: Batch, steps: int = 48, rounds: int = 2, k1_forms: Optional[bool] = None) -> dict:
        """Time the step's equivalent schedules on THIS device with THIS batch shape, in the regime the trainer will run
        in, and keep the fastest: phase B's two backward schedules (include/tt_hotpath.h, tt_engine_set_backward), where
        the gather's slot fence comes from (single replica), and — with a gradient exchange whose form the caller left
        open (`split_exchange=None`) — the exchange in one bucket or in two parts.  Single replica: dry steps with the
        gathers running beside the chain exactly as in training.  Data parallel (round 5): the dry steps ARE the
        data-parallel step — slab reduce into a scratch bucket, the next gather issued under the exchange, the real
        collective on the caller's stream — because there the chain runs alone on the chip and the ranking may differ
        from the one beside the gathers (configs[2]'s share under a 60 us stand-in: the two tie at 186 us without a
        profiler, fused is 4.6 us ahead under rocprofv3; docs/LOG_r05.md section 1).  All
        candidates write bitwise the same gradients, so the choice never changes a training run's parameters.
        Parameters, Adam state, step counters, the gradient bucket and loss statistics are untouched (no optimizer).
        A side effect worth knowing: the dry steps also bring the GPU from its idle power state to sustained clocks,
        which otherwise takes the first ~70 training steps (docs/LOG_r04.md §5 "Cold start").
        `k1_forms` adds the gather's form as a candidate where the shape rule picks the binned form: that rule assumes
        every table row is named about equally often, and on THIS batch's ids the streaming form may be the faster one
        (log-uniform ids at configs[3]: 4.4 vs 5.1 ms per step, profiles/r05_c4_skew.txt).  None (default, round 6): on for
        a table beyond the Infinity Cache — the only tables the binned form is picked for — unless `tuning` fixes a form.
        Unlike the other candidates the two forms differ in fp32 summation order (both within the 1e-5 parity bound, each
        bitwise run to run), so the choice is recorded — `calibration["k1_form"]`, and `state_dict()["k1_form"]`, which
        `load_state_dict` re-applies so that a resumed run keeps its form; a run that must be bitwise repeatable ACROSS
        calibrations fixes the form in `tuning`.  Data parallel: the candidate set is agreed between the ranks first (a
        rank whose shard is too short to bin would otherwise run a different number of collectives)."""
        if self._pending:
            raise RuntimeError("calibrate(): call it between epochs (batches are still in flight)")
        if self.device.type != "cuda" or not hasattr(self.engine, "set_backward") or self.loss_kind != "triplet":
            return {}
        b = as_batch(batch)
        depth = self.SLOTS - 1 if self.side is not None else 1
        # second dimension: where the gather's slot fence comes from (see __init__, fence_at_loss) — only where the
        # caller left it open and the engine's loss event exists
        fences = (False, True) if (self.side is not None and self._loss_event is not None and self._fence_arg is None) else (self._use_loss_fence,)
        # third: the exchange in one bucket or two parts — only where the caller left it open
        forms = (False, True) if (self._exchange and self._split_auto) else (self._split,)
        base_tuning = dict(self.engine.tuning or {}) if hasattr(self.engine, "tuning") else {}
        gathers = (None,)
        if k1_forms is None:
            k1_forms = (hasattr(self.engine, "V") and self.engine.V * self.D * 4 > self.INFINITY_CACHE_BYTES
                        and getattr(self, "_k1_form_loaded", None) is None)
        ids_like = b.ids if b.ids is not None else b.token_rows
        width = (b.ids.shape[2] if b.ids is not None else
                 (b.token_store if b.token_store is not None else (self.token_store or (None,))[0]))
        if width is not None and not isinstance(width, int):
            width = width.shape[1]
        want = bool(k1_forms and hasattr(self.engine, "set_tuning") and not base_tuning.get("k1_form") and ids_like is not None
                    and width is not None and b.size > 0
                    and _lib.load().tt_k1_prefers_binned(self.engine.V, self.D, ids_like.shape[0] * b.size, width))
        if self.world > 1 and k1_forms and hasattr(self.engine, "set_tuning"):
            agree = torch.tensor([int(want)], dtype=torch.int32, device=self.device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=self.pg)  # every rank times the same candidates, or none does
            want = bool(int(agree.item()))
        if want:
            gathers = (None, "stream")
        times = {(m, f, sp, kf): [] for m in ("fused", "fork") for f in fences for sp in forms for kf in gathers}
        saved = (self._use_loss_fence, self._split, getattr(self.engine, "backward_mode", None))
        self._dry = True
        if self._exchange and self.bucket is not None:
            self._dry_bucket = torch.zeros_like(self.bucket)  # the real bucket may hold an un-applied accumulation window
        ok = False
        try:
            for _ in range(rounds):
                for mode, fence, sp, kf in times:
                    self._use_loss_fence = fence
                    self._split = sp
                    if len(gathers) > 1:
                        self.engine.set_tuning({**base_tuning, **({"k1_form": kf} if kf else {})})
                    self.engine.set_backward(mode)
                    while len(self._pending) < depth:
                        self.gather(b)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(steps):
                        self.apply(None, issue_next=lambda fence=None: self.gather(b, fence=fence))
                    e1.record()
                    while self._pending:
                        self.apply(None)
                    e1.synchronize()
                    times[(mode, fence, sp, kf)].append(e0.elapsed_time(e1) / steps)
            ok = True
        finally:
            if not ok:  # an error mid-calibration: no candidate's settings and no dry batch may stay behind
                try:
                    if self._pending:
                        self.abort_pipeline()
                finally:
                    self._use_loss_fence, self._split = saved[0], saved[1]
                    if len(gathers) > 1:
                        self.engine.set_tuning(base_tuning)
                    if saved[2] is not None:
                        self.engine.set_backward(saved[2])
            self._dry = False
            self._dry_bucket = None
        keys = list(times)
        best = {kf: min(v) for kf, v in times.items()}
        if self.world > 1:  # every replica must take the same decision: decide on the slowest rank's times
            t = torch.tensor([best[kf] for kf in keys], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
            best = {kf: float(t[i]) for i, kf in enumerate(keys)}
        pick = min(best, key=best.get)
        if len(gathers) > 1:
            self.engine.set_tuning({**base_tuning, **({"k1_form": pick[3]} if pick[3] else {})})
        self.engine.set_backward(pick[0])
        self._use_loss_fence = pick[1]
        self._split = pick[2]
        self.engine.read_stats(reset=True)  # the dry steps' losses do not belong to any epoch
        self.calibration = {"backward": pick[0], "fence_at_loss": pick[1],
                            "exchange_form": (None if not self._exchange else ("two-part" if pick[2] else "one-bucket")),
                            "regime": ("data-parallel step: slab reduce + exchange inside the timed dry steps, the gather under "
                                       "the exchange" if self._exchange else "single replica: gathers beside the chain"),
                            "ms_per_step": {m: min(best[kf] for kf in keys if kf[0] == m) for m in ("fused", "fork")},
                            "k1_form": (pick[3] or "shape rule") if len(gathers) > 1 else getattr(self, "_k1_form_loaded", None),
                            "ms_per_step_by_fence": {f"{m},{'loss' if f else 'own'}" + (",two-part" if sp else "") + (f",k1={kf}" if kf else ""):
                                                     best[(m, f, sp, kf)] for m, f, sp, kf in keys},
                            "steps": steps, "rounds": rounds}
        return self.calibration
