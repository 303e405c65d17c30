"""Optimiser and data-parallel plumbing of the step: one flat fp32 parameter/gradient buffer, a fused Adam
kernel over it (dis_adam_step) and ONE gradient all-reduce per step over RCCL (torch.distributed 'nccl').

The reference uses torch.optim.Adam(lr=1e-4) on a single GPU (train_val.py:55-56) and has no communication
layer; data parallelism here is batch sharding with mean-of-per-rank gradients."""
import torch

from . import ops


class FlatAdam(object):
    """Adam with torch.optim.Adam default hyper-parameters on a flattened view of `params`.

    All parameters (also the never-used `upconv1/2` of FuseNet, whose gradient stays zero: a zero gradient
    leaves an Adam parameter unchanged) are re-pointed into one contiguous buffer; gradients are accumulated by
    autograd directly into the matching views of one flat gradient buffer."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, world_size=1, process_group=None):
        self.params = [p for p in params]
        assert len(self.params) > 0
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        pad = (n + 3) // 4 * 4
        self.flat_p = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(pad, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(pad, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                p.grad = self.flat_g[off:off + k].view(p.shape)
                off += k
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.world_size = world_size
        self.process_group = process_group
        if dev.type == 'cuda':
            ops.register_grad_sinks(self.params)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        ops.reset_grad_sinks()
        if self.flat_g.is_cuda:
            ops.begin_step(self.flat_g.device)

    def all_reduce_grads(self):
        if self.world_size > 1:
            torch.distributed.all_reduce(self.flat_g, group=self.process_group)

    def step(self, all_reduce=True):
        if all_reduce:
            self.all_reduce_grads()
        self.step_count += 1
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                      self.betas[0], self.betas[1], self.eps, 1.0 / self.world_size)

    # torch.optim-compatible (de)serialisation so Worker.train can checkpoint it
    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg[:self.n].clone(),
                'exp_avg_sq': self.exp_avg_sq[:self.n].clone(), 'lr': self.lr, 'betas': self.betas, 'eps': self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg[:self.n].copy_(sd['exp_avg'])
        self.exp_avg_sq[:self.n].copy_(sd['exp_avg_sq'])
