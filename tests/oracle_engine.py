"""TEST INFRASTRUCTURE: a CPU stand-in for HotPathEngine backed by the oracle, so
the data-parallel / accumulation / prefetch logic of TripletTrainer can be
exercised with gloo on CPU.  Never imported by the product package."""
import torch

from oracle import two_towers_oracle as orc
from two_towers_overlords_amd.engine import flat_param_slices, tower_param_count


class OracleEngine:
    def __init__(self, table, projection_dim, max_batch, max_seq_len):
        self.table = table
        self.V, self.D = table.shape
        self.P = projection_dim
        self.n_params = 2 * tower_param_count(self.D, self.P)
        self.n_slabs = 1
        self.pooled = [None, None, None]
        self.loss_sum, self.count, self.last = 0.0, 0, 0.0
        self.grad = torch.zeros(self.n_params)

    def bind(self, params, exp_avg, exp_avg_sq):
        self.params, self.m, self.v = params, exp_avg, exp_avg_sq

    def _named(self, flat):
        out = {}
        for name, off, shape in flat_param_slices(self.D, self.P):
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off: off + n].view(shape)
        return out

    def pool(self, slot, ids, mask=None, lens=None, stream=None):
        S = ids.shape[2]
        if mask is None:
            if lens is None:
                mask = torch.ones_like(ids, dtype=torch.int64)
            else:
                mask = (torch.arange(S).view(1, 1, S) < lens.unsqueeze(-1)).long()
        self.pooled[slot] = [orc.gather_pool_normalize(self.table, ids[r].long(), mask[r].long()) for r in range(ids.shape[0])]

    def pool_cached(self, slot, cache, rows, stream=None):
        self.pooled[slot] = [cache[rows[r].long()] for r in range(rows.shape[0])]

    def fwd_bwd(self, slot, B, margin, grad_scale=None, loss_denominator=None, neg_index=None):
        gs = 1.0 / B if grad_scale is None else grad_scale
        den = float(B) if loss_denominator is None else loss_denominator
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in self._named(self.params).items()}
        embs = []
        towers = ("query_tower", "document_tower", "document_tower")[: len(self.pooled[slot])]
        for r, tower in enumerate(towers):
            k = tower + ".projection."
            embs.append(orc.projection(self.pooled[slot][r], leaves[k + "0.weight"], leaves[k + "0.bias"],
                                       leaves[k + "2.weight"], leaves[k + "2.bias"]))
        if neg_index is not None:
            embs.append(embs[1][neg_index.long()])  # negatives are in-batch positives
        pd = orc.cosine_distance(embs[0], embs[1])
        nd = orc.cosine_distance(embs[0], embs[2])
        hinge_sum = torch.relu(pd - nd + margin).sum()
        (hinge_sum * gs).backward()
        self.grad = torch.cat([leaves[name].grad.reshape(-1) for name, _, _ in flat_param_slices(self.D, self.P)])
        self.last = float(hinge_sum.detach()) / den
        self.loss_sum += self.last
        self.count += 1

    def reduce_grads(self, out, accumulate=False):
        if accumulate:
            out += self.grad
        else:
            out.copy_(self.grad)

    def reduce_grads_range(self, out, offset, count, accumulate=False):
        if accumulate:
            out[offset: offset + count] += self.grad[offset: offset + count]
        else:
            out[offset: offset + count].copy_(self.grad[offset: offset + count])

    def adam(self, step, lr, grads=None, beta1=0.9, beta2=0.999, eps=1e-8):
        g = self.grad if grads is None else grads
        st = orc.AdamState(lr=lr, beta1=beta1, beta2=beta2, eps=eps, step=step - 1, m={"p": self.m}, v={"p": self.v})
        orc.adam_step({"p": self.params}, {"p": g}, st)

    def read_stats(self, reset=True):
        out = (self.loss_sum, self.count, self.last)
        if reset:
            self.loss_sum, self.count = 0.0, 0
        return out
