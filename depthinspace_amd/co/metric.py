"""Evaluation metrics of `retest` / `test_epoch` (reference co/metric.py:60-154, used by the stage workers'
callback_test_* hooks, model/multi_frame_worker.py:236-263): DistanceMetric + OutlierFractionMetric over all test pixels,
kept ON THE DEVICE (the reference copies every output to the host and keeps every distance in numpy lists).

Same class names, constructor arguments, add() / get() / items() / __str__ and result keys (`dist2_mean`, `dist2_std`,
`dist2_median`, `dist2_q10`, `dist2_q90`, `dist2_min`, `dist2_max`, `of0.1` ... `of5`).  add() takes torch tensors of shape
(N, vec_length); nothing is synchronised until get().  Order statistics follow numpy (np.median, np.percentile with linear
interpolation); moments are accumulated in float64.  Under data parallelism every rank adds its shard and get() merges
the ranks (all_gather), so every rank returns the metrics of the whole test set.
"""
import torch


def _dist_group():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def _fallback_device():
    """where a rank that never saw a sample (empty test shard) builds its empty contribution: the device its peers use"""
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def _gather_cat(x):
    """concatenation of a 1-D tensor over all ranks (sizes may differ)"""
    if not _dist_group():
        return x
    world = torch.distributed.get_world_size()
    n = torch.tensor([x.numel()], device=x.device, dtype=torch.int64)
    ns = [torch.zeros_like(n) for _ in range(world)]
    torch.distributed.all_gather(ns, n)
    m = int(max(int(v) for v in ns))
    buf = torch.zeros(m, device=x.device, dtype=x.dtype)
    buf[:x.numel()] = x
    outs = [torch.zeros_like(buf) for _ in range(world)]
    torch.distributed.all_gather(outs, buf)
    return torch.cat([o[:int(k)] for o, k in zip(outs, ns)])


class Metric(object):
    def __init__(self, str_prefix=''):
        self.str_prefix = str_prefix
        self.reset()

    def reset(self):
        pass

    def add(self, es, ta, ma=None):
        pass

    def get(self):
        return {}

    def items(self):
        return self.get().items()

    def __str__(self):
        return ', '.join([f'{self.str_prefix}{key}={value:.5f}' for key, value in self.get().items()])


class MultipleMetric(Metric):
    def __init__(self, *metrics, **kwargs):
        self.metrics = [*metrics]
        super().__init__(**kwargs)

    def reset(self):
        for m in self.metrics:
            m.reset()

    def add(self, es, ta, ma=None):
        for m in self.metrics:
            m.add(es, ta, ma)

    def get(self):
        ret = {}
        for m in self.metrics:
            ret.update(m.get())
        return ret

    def __str__(self):
        return '\n'.join([str(m) for m in self.metrics])


def _distances(es, ta, ma, vec_length, p):
    if es.shape != ta.shape or es.dim() != 2 or es.shape[1] != vec_length:
        raise Exception('es and ta have to be of shape Nxdim')
    d = es - ta
    dist = d.abs()[:, 0] if vec_length == 1 else torch.linalg.vector_norm(d, ord=p, dim=1)
    if ma is not None:
        dist = dist[ma.reshape(-1) != 0]
    return dist


def _percentile_sorted(s, q):
    """np.percentile(..., q) (linear interpolation) of an ascending 1-D tensor"""
    n = s.numel()
    pos = (n - 1) * (q / 100.0)
    lo = int(pos)
    hi = min(lo + 1, n - 1)
    frac = pos - lo
    a, b = float(s[lo]), float(s[hi])
    return a + (b - a) * frac


class DistanceMetric(Metric):
    """reference co/metric.py:104-137"""

    def __init__(self, vec_length, p=2, **kwargs):
        self.vec_length = vec_length
        self.p = p
        self.name = f'{p}'
        super().__init__(**kwargs)

    def reset(self):
        self.dists = []

    def add(self, es, ta, ma=None):
        self.dists.append(_distances(es, ta, ma, self.vec_length, self.p))

    def _all(self):
        # a rank without samples (fewer test tracks than ranks) still takes part in the collectives
        mine = torch.cat(self.dists) if self.dists else torch.empty(0, dtype=torch.float32, device=_fallback_device())
        return _gather_cat(mine)

    def get(self):
        d = self._all()
        n = d.numel()
        if n == 0:   # nothing was added on any rank
            nan = float('nan')
            return {f'dist{self.name}_{k}': nan for k in ('mean', 'std', 'median', 'q10', 'q90', 'min', 'max')}
        s, _ = torch.sort(d)
        d64 = d.double()
        mean = float(d64.sum()) / n
        var = float(((d64 - mean) ** 2).sum()) / n
        return {
            f'dist{self.name}_mean': mean,
            f'dist{self.name}_std': var ** 0.5,
            f'dist{self.name}_median': _percentile_sorted(s, 50.0),
            f'dist{self.name}_q10': _percentile_sorted(s, 10.0),
            f'dist{self.name}_q90': _percentile_sorted(s, 90.0),
            f'dist{self.name}_min': float(s[0]),
            f'dist{self.name}_max': float(s[-1]),
        }


class OutlierFractionMetric(DistanceMetric):
    """reference co/metric.py:139-154; only exact integer counts are kept"""

    def __init__(self, thresholds, *args, **kwargs):
        self.thresholds = thresholds
        super().__init__(*args, **kwargs)

    def reset(self):
        self.counts = None
        self.total = 0

    def add(self, es, ta, ma=None):
        dist = _distances(es, ta, ma, self.vec_length, self.p)
        c = torch.stack([(dist > t).sum() for t in self.thresholds])
        self.counts = c if self.counts is None else self.counts + c
        self.total += dist.numel()

    def get(self):
        if self.counts is None:   # empty shard: zero counts, same collectives as everyone else
            c = torch.zeros(len(self.thresholds), dtype=torch.int64, device=_fallback_device())
        else:
            c = self.counts.clone()
        tot = torch.tensor([self.total], device=c.device, dtype=torch.int64)
        if _dist_group():
            torch.distributed.all_reduce(c)
            torch.distributed.all_reduce(tot)
        n = int(tot)
        return {f'of{t}': (float(int(v)) / float(n) if n > 0 else float('nan')) for t, v in zip(self.thresholds, c)}
